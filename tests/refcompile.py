"""Compile-only check of the Fortran drop-in against the REFERENCE'S OWN module interfaces (test infrastructure).

What it is: the module closure of `source/ice_step_mod.F90` is compiled for its .mod files from the unmodified reference
sources where they lie under /root/reference (nothing is copied into this repository, every output goes to a scratch
directory), then OUR `fortran/evpk_mod.F90` and `fortran/ice_dyn_evp.F90` are compiled against those .mod files, and
finally the reference's unmodified `source/ice_step_mod.F90` against OUR `ice_dyn_evp.mod` -- the `use ice_dyn_evp, only:
evp` / `call evp (dt)` of ice_step_mod.F90:1085,1119.  That checks the shim's `use ... only:` lists, kinds, ranks and
argument lists against the real entities instead of the hand-written test doubles of fortran/mock/.

What it is NOT: a reference build.  Nothing is linked or run, no number comes out of it, nothing it produces is an
oracle or a fixture, and it never ships.  The one module the image lacks, `netcdf` (used unconditionally by
source/ice_grid.F90:144,254,887), is given as an INTERFACE-ONLY declaration below so that the `use netcdf` statements
resolve; its procedures have no bodies worth the name and are never called.

Recipe: search path <stub>, drivers/auscom, source, serial, io_binary, csm_share (first file of a name
wins, as on the reference's VPATH); `ice_grid`'s closure is compiled WITHOUT -DAusCOM (io code behind that define needs
a real netCDF), `ice_constants` and everything else WITH it; dependencies from the `use` statements of the preprocessed
sources.
"""
from __future__ import annotations

import os
import re
import subprocess
from typing import Dict, List, Set

REF = "/root/reference"
FC = "/opt/rocm/bin/amdflang"
VPATH = ["drivers/auscom", "source", "serial", "io_binary", "csm_share"]
DEFS = ("-DLINUX -DNXGLOB=100 -DNYGLOB=116 -DBLCKX=25 -DBLCKY=29 -DMXBLCKS=16 -DNICELYR=4 -DNSNWLYR=1 -DNICECAT=5 "
        "-DTRAGE=1 -DTRFY=1 -DTRLVL=1 -DTRPND=1 -DTRBRI=1 -DNTRAERO=0 -DNBGCLYR=0 -DTRBGCS=2 -DNUMIN=11 -DNUMAX=99").split()

NETCDF_STUB = """\
! interface-only stand-in so that `use netcdf` resolves (tests/refcompile.py: compile-only check, never linked or run)
module netcdf
   implicit none
   integer, parameter :: nf90_noerr = 0, nf90_nowrite = 0, nf90_write = 1, nf90_clobber = 0, nf90_global = 0, &
                         nf90_double = 6, nf90_float = 5, nf90_int = 4, nf90_char = 2, nf90_unlimited = 0, nf90_64bit_offset = 512
   interface nf90_get_var
      module procedure gv_r8_0, gv_r8_1, gv_r8_2, gv_r8_3, gv_i4_0, gv_i4_1, gv_i4_2
   end interface
   interface nf90_put_var
      module procedure pv_r8_0, pv_r8_1, pv_r8_2, pv_r8_3, pv_r4_2, pv_r4_3, pv_i4_0
   end interface
   interface nf90_put_att
      module procedure pa_c, pa_r8, pa_r4, pa_i4
   end interface
   interface nf90_get_att
      module procedure ga_c, ga_r8
   end interface
contains
   integer function nf90_open(path, mode, ncid)
      character(*), intent(in) :: path; integer, intent(in) :: mode; integer, intent(out) :: ncid
      ncid = -1; nf90_open = -1
   end function
   integer function nf90_create(path, cmode, ncid)
      character(*), intent(in) :: path; integer, intent(in) :: cmode; integer, intent(out) :: ncid
      ncid = -1; nf90_create = -1
   end function
   integer function nf90_close(ncid)
      integer, intent(in) :: ncid
      nf90_close = -1
   end function
   integer function nf90_enddef(ncid)
      integer, intent(in) :: ncid
      nf90_enddef = -1
   end function
   integer function nf90_redef(ncid)
      integer, intent(in) :: ncid
      nf90_redef = -1
   end function
   integer function nf90_sync(ncid)
      integer, intent(in) :: ncid
      nf90_sync = -1
   end function
   integer function nf90_inq_varid(ncid, name, varid)
      integer, intent(in) :: ncid; character(*), intent(in) :: name; integer, intent(out) :: varid
      varid = -1; nf90_inq_varid = -1
   end function
   integer function nf90_inq_dimid(ncid, name, dimid)
      integer, intent(in) :: ncid; character(*), intent(in) :: name; integer, intent(out) :: dimid
      dimid = -1; nf90_inq_dimid = -1
   end function
   integer function nf90_inquire_dimension(ncid, dimid, name, len)
      integer, intent(in) :: ncid, dimid; character(*), intent(out), optional :: name; integer, intent(out), optional :: len
      if (present(name)) name = ' '
      if (present(len)) len = 0
      nf90_inquire_dimension = -1
   end function
   integer function nf90_inquire_variable(ncid, varid, name, xtype, ndims, dimids, natts)
      integer, intent(in) :: ncid, varid; character(*), intent(out), optional :: name
      integer, intent(out), optional :: xtype, ndims, natts; integer, intent(out), optional :: dimids(:)
      if (present(name)) name = ' '
      if (present(xtype)) xtype = 0
      if (present(ndims)) ndims = 0
      if (present(natts)) natts = 0
      if (present(dimids)) dimids = 0
      nf90_inquire_variable = -1
   end function
   integer function nf90_inquire(ncid, ndimensions, nvariables, nattributes, unlimiteddimid)
      integer, intent(in) :: ncid; integer, intent(out), optional :: ndimensions, nvariables, nattributes, unlimiteddimid
      if (present(ndimensions)) ndimensions = 0
      if (present(nvariables)) nvariables = 0
      if (present(nattributes)) nattributes = 0
      if (present(unlimiteddimid)) unlimiteddimid = 0
      nf90_inquire = -1
   end function
   integer function nf90_def_dim(ncid, name, len, dimid)
      integer, intent(in) :: ncid, len; character(*), intent(in) :: name; integer, intent(out) :: dimid
      dimid = -1; nf90_def_dim = -1
   end function
   integer function nf90_def_var(ncid, name, xtype, dimids, varid)
      integer, intent(in) :: ncid, xtype; character(*), intent(in) :: name; integer, intent(in), optional :: dimids(..)
      integer, intent(out) :: varid
      varid = -1; nf90_def_var = -1
   end function
   function nf90_strerror(ncerr)
      integer, intent(in) :: ncerr; character(80) :: nf90_strerror
      nf90_strerror = 'netcdf is not part of this compile-only check'
   end function
#define GV(NAME, T, DIMS) \\
   integer function NAME(ncid, varid, values, start, count); \\
      integer, intent(in) :: ncid, varid; T, intent(out) :: values DIMS; integer, intent(in), optional :: start(:), count(:); \\
      values = 0; NAME = -1; \\
   end function
#define PV(NAME, T, DIMS) \\
   integer function NAME(ncid, varid, values, start, count); \\
      integer, intent(in) :: ncid, varid; T, intent(in) :: values DIMS; integer, intent(in), optional :: start(:), count(:); \\
      NAME = -1; \\
   end function
   GV(gv_r8_0, real(8), )
   GV(gv_r8_1, real(8), (:))
   GV(gv_r8_2, real(8), (:,:))
   GV(gv_r8_3, real(8), (:,:,:))
   GV(gv_i4_0, integer, )
   GV(gv_i4_1, integer, (:))
   GV(gv_i4_2, integer, (:,:))
   PV(pv_r8_0, real(8), )
   PV(pv_r8_1, real(8), (:))
   PV(pv_r8_2, real(8), (:,:))
   PV(pv_r8_3, real(8), (:,:,:))
   PV(pv_r4_2, real(4), (:,:))
   PV(pv_r4_3, real(4), (:,:,:))
   PV(pv_i4_0, integer, )
   integer function pa_c(ncid, varid, name, values)
      integer, intent(in) :: ncid, varid; character(*), intent(in) :: name, values
      pa_c = -1
   end function
   integer function pa_r8(ncid, varid, name, values)
      integer, intent(in) :: ncid, varid; character(*), intent(in) :: name; real(8), intent(in) :: values
      pa_r8 = -1
   end function
   integer function pa_r4(ncid, varid, name, values)
      integer, intent(in) :: ncid, varid; character(*), intent(in) :: name; real(4), intent(in) :: values
      pa_r4 = -1
   end function
   integer function pa_i4(ncid, varid, name, values)
      integer, intent(in) :: ncid, varid; character(*), intent(in) :: name; integer, intent(in) :: values
      pa_i4 = -1
   end function
   integer function ga_c(ncid, varid, name, values)
      integer, intent(in) :: ncid, varid; character(*), intent(in) :: name; character(*), intent(out) :: values
      values = ' '; ga_c = -1
   end function
   integer function ga_r8(ncid, varid, name, values)
      integer, intent(in) :: ncid, varid; character(*), intent(in) :: name; real(8), intent(out) :: values
      values = 0; ga_r8 = -1
   end function
end module netcdf
"""

_USE = re.compile(r"^\s*use\s+(\w+)", re.I | re.M)
_MOD = re.compile(r"^\s*module\s+(?!procedure\b)(\w+)", re.I | re.M)


class RefCompile:
    def __init__(self, work: str, extra_defs: List[str] = ()):
        """extra_defs: further cpp defines for the files compiled with -DAusCOM (e.g. ["-DACCESS"])"""
        self.work = work
        self.extra = list(extra_defs)
        self.mods = os.path.join(work, "mods")
        os.makedirs(self.mods, exist_ok=True)
        self.stub = os.path.join(work, "netcdf_stub.F90")
        open(self.stub, "w").write(NETCDF_STUB)
        # first file of a name on the search path wins
        self.files: Dict[str, str] = {}
        for d in VPATH:
            p = os.path.join(REF, d)
            for n in sorted(os.listdir(p)):
                if n.endswith(".F90") and n not in self.files:
                    self.files[n] = os.path.join(p, n)
        self.pp: Dict[tuple, str] = {}
        self.module_file: Dict[str, str] = {}
        for n, path in self.files.items():
            for m in _MOD.findall(open(path, errors="replace").read()):
                self.module_file.setdefault(m.lower(), path)
        self.done: Set[str] = set()
        self.log: List[str] = []

    def _cpp(self, path: str, auscom: bool) -> str:
        key = (path, auscom)
        if key not in self.pp:
            cmd = ["cpp", "-P", "-traditional"] + DEFS + (["-DAusCOM"] + self.extra if auscom else []) + [path]
            self.pp[key] = subprocess.run(cmd, capture_output=True, text=True, errors="replace").stdout
        return self.pp[key]

    def uses(self, path: str, auscom: bool) -> List[str]:
        return sorted({m.lower() for m in _USE.findall(self._cpp(path, auscom))})

    def closure(self, module: str, auscom: bool, skip: Set[str] = frozenset()) -> List[str]:
        """files `module` needs, dependencies first"""
        order: List[str] = []
        seen: Set[str] = set()

        def visit(m):
            path = self.module_file.get(m)
            if path is None or path in seen or m in skip:
                return
            seen.add(path)
            for u in self.uses(path, auscom):
                visit(u)
            order.append(path)

        visit(module.lower())
        return order

    def compile(self, path: str, auscom: bool, extra: List[str] = ()):
        cmd = [FC, "-cpp", "-fsyntax-only"] + DEFS + (["-DAusCOM"] + self.extra if auscom else []) + list(extra) + \
              ["-module-dir", self.mods, "-I", self.mods, "-I", os.path.join(REF, "drivers/auscom"), path]
        r = subprocess.run(cmd, capture_output=True, text=True, errors="replace")
        self.log.append(("ok  " if r.returncode == 0 else "FAIL") + " " + os.path.relpath(path, REF if path.startswith(REF) else self.work)
                        + ("  [AusCOM]" if auscom else ""))
        if r.returncode != 0:
            raise RuntimeError(f"{path}:\n{r.stderr[-4000:]}")

    def need(self, top: str, skip: Set[str] = frozenset(), include_top: bool = True):
        """compile (for its .mod files) everything module `top` uses, dependencies first, and `top` itself if asked;
        `skip`: modules that are provided otherwise (ours)"""
        if not self.done:
            self.compile(self.stub, False)
            self.done.add(self.stub)
            self.grid = set(self.closure("ice_grid", False, skip={"ice_constants"}))
            for path in self.closure("ice_constants", True):
                self.compile(path, True); self.done.add(path)
            for path in self.closure("ice_grid", False):
                if path not in self.done:
                    self.compile(path, False); self.done.add(path)
        topfile = self.module_file[top.lower()]
        for path in self.closure(top, True, skip=set(skip)):
            if path in self.done or (path == topfile and not include_top):
                continue
            self.compile(path, path not in self.grid); self.done.add(path)
