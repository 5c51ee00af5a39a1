"""The Fortran drop-in against the reference's REAL module interfaces -- compile only, CPU, skipped where /root/reference
or amdflang is absent (the GPU box).  See tests/refcompile.py for what this is and is not: it builds .mod files from the
unmodified reference sources in a scratch directory to check OUR shim's `use` lists, kinds, ranks and argument lists;
nothing is linked or run, nothing it produces is an oracle, nothing ships."""
import os
import subprocess

import pytest

from tests import refcompile as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not (os.path.isdir(R.REF) and os.path.exists(R.FC)),
                                reason="needs /root/reference and amdflang (this container only)")


@pytest.mark.parametrize("extra", [[], ["-DACCESS"]], ids=["AusCOM", "AusCOM+ACCESS"])
def test_shim_compiles_against_reference_modules_and_ice_step_mod_against_the_shim(tmp_path, extra):
    """source/ice_step_mod.F90:1085 `use ice_dyn_evp, only: evp`, :1119 `if (kdyn == 1) call evp (dt)` -- unmodified --
    compiles against OUR ice_dyn_evp.mod; our module compiles against the reference's ice_state / ice_flux / ice_grid /
    ice_dyn_shared / ice_mechred / ... as they are (source/ice_dyn_evp.F90:70-96)."""
    rc = R.RefCompile(str(tmp_path), extra)
    rc.need("ice_dyn_evp", include_top=False)                 # what the reference's own ice_dyn_evp needs
    ref_evp = os.path.join(R.REF, "source", "ice_dyn_evp.F90")
    assert ref_evp not in rc.done
    rc.compile(os.path.join(ROOT, "fortran", "evpk_mod.F90"), True)
    rc.compile(os.path.join(ROOT, "fortran", "ice_dyn_evp.F90"), True)
    assert os.path.exists(os.path.join(rc.mods, "ice_dyn_evp.mod"))
    rc.need("ice_step_mod", skip={"ice_dyn_evp"})             # ... the rest of the model, then ice_step_mod itself
    assert os.path.join(R.REF, "source", "ice_step_mod.F90") in rc.done
    assert os.path.join(R.REF, "source", "ice_dyn_eap.F90") in rc.done        # shares evp_prep1/2, stepu of ice_dyn_shared
    assert ref_evp not in rc.done                             # the reference's own module was never compiled: ours stood in


@pytest.mark.parametrize("extra", [[], ["-DACCESS"]], ids=["AusCOM", "AusCOM+ACCESS"])
def test_mock_modules_declare_every_imported_entity_as_the_reference_does(tmp_path, extra):
    """tests/fortran/entities_tkr.F90 passes every entity the shim imports to a dummy of one type, kind and rank and calls
    every imported procedure with the shim's argument list: it must compile against the real modules AND against
    fortran/mock/cice_mock_modules.F90."""
    src = os.path.join(ROOT, "tests", "fortran", "entities_tkr.F90")
    rc = R.RefCompile(str(tmp_path / "ref"), extra)
    rc.need("ice_dyn_evp", include_top=False)
    rc.compile(src, True)
    mock = tmp_path / "mock"
    mock.mkdir()
    for f in (os.path.join(ROOT, "fortran", "mock", "cice_mock_modules.F90"), src):
        r = subprocess.run([R.FC, "-cpp", "-fsyntax-only", "-DAusCOM"] + extra + ["-module-dir", str(mock), "-I", str(mock), f],
                           capture_output=True, text=True)
        assert r.returncode == 0, f + "\n" + r.stderr[-3000:]


def test_transport_driver_compiles_against_the_shims_horizontal_remap(tmp_path):
    """source/ice_transport_driver.F90 with ONE line changed -- :216 `use ice_transport_remap, only: horizontal_remap,
    make_masks` -> make_masks from there, horizontal_remap => evpk_horizontal_remap from our ice_dyn_evp -- compiles: the call
    at :475-481 (aim, trm, tracer_type, depend, has_dependents, integral_order, l_dp_midpt as the driver declares them)
    matches the shim's argument list.  The edited text exists in the scratch directory only."""
    rc = R.RefCompile(str(tmp_path), [])
    rc.need("ice_dyn_evp", include_top=False)
    rc.compile(os.path.join(ROOT, "fortran", "evpk_mod.F90"), True)
    rc.compile(os.path.join(ROOT, "fortran", "ice_dyn_evp.F90"), True)
    rc.need("ice_transport_driver", include_top=False, skip={"ice_dyn_evp"})
    src = open(os.path.join(R.REF, "source", "ice_transport_driver.F90")).read()
    old = "      use ice_transport_remap, only: horizontal_remap, make_masks\n"
    assert src.count(old) == 1
    edited = tmp_path / "ice_transport_driver_edited.F90"
    edited.write_text(src.replace(old, "      use ice_transport_remap, only: make_masks\n"
                                       "      use ice_dyn_evp, only: horizontal_remap => evpk_horizontal_remap\n"))
    rc.compile(str(edited), True)
    assert os.path.exists(os.path.join(rc.mods, "ice_transport_driver.mod"))


EAP_BODY = """      subroutine eap (dt)
      use ice_dyn_evp, only: evpk_eap
      real (kind=dbl_kind), intent(in) :: dt
      call evpk_eap (dt, nx_yield, ny_yield, na_yield, s11r, s12r, s22r, s11s, s12s, s22s, &
                     a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12,    &
                     e11, e12, e22, yieldstress11, yieldstress12, yieldstress22, s11, s12, s22)
      end subroutine eap
"""


def test_reference_eap_module_compiles_with_its_eap_body_replaced_by_the_shim_call(tmp_path):
    """source/ice_dyn_eap.F90 with the body of `subroutine eap (dt)` (:66-486) replaced by the call INTEGRATION.md shows:
    the module's own private tables and structure tensor match evpk_eap's dummies; ice_step_mod.F90 (`use ice_dyn_eap, only:
    eap`, :1084) then compiles against the edited module.  The edited text exists in the scratch directory only."""
    rc = R.RefCompile(str(tmp_path), [])
    rc.need("ice_dyn_evp", include_top=False)
    rc.compile(os.path.join(ROOT, "fortran", "evpk_mod.F90"), True)
    rc.compile(os.path.join(ROOT, "fortran", "ice_dyn_evp.F90"), True)
    rc.need("ice_dyn_eap", include_top=False, skip={"ice_dyn_evp"})
    lines = open(os.path.join(R.REF, "source", "ice_dyn_eap.F90")).read().split("\n")
    a = next(k for k, l in enumerate(lines) if l.strip().lower().startswith("subroutine eap (dt)"))
    b = next(k for k, l in enumerate(lines) if l.strip().lower().startswith("end subroutine eap"))
    assert 60 < a < 70 and 480 < b < 490
    edited = tmp_path / "ice_dyn_eap_edited.F90"
    edited.write_text("\n".join(lines[:a]) + "\n" + EAP_BODY + "\n".join(lines[b + 1:]) + "\n")
    rc.compile(str(edited), True)
    assert os.path.exists(os.path.join(rc.mods, "ice_dyn_eap.mod"))
    rc.need("ice_step_mod", skip={"ice_dyn_evp", "ice_dyn_eap"})
    assert os.path.join(R.REF, "source", "ice_step_mod.F90") in rc.done

