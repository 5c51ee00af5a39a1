"""Host-side mirror of the table part of init_eap (source/ice_dyn_eap.F90:555-619, functions w1, w2, s11kr ... s22ks :626-1046):
the six lookup tables of the EAP rheology, with numpy.  They are an INPUT of libevpk (evpk_eap_init) -- a Fortran host passes
the module arrays its own init_eap filled; this is the same step for the Python mirror (dyn.EvpDynamics.init_eap).  Returned as
C arrays [na_yield][ny_yield][nx_yield], i.e. the memory layout of the Fortran s11r(nx_yield, ny_yield, na_yield)."""
import numpy as np

NX_YIELD, NY_YIELD, NA_YIELD = 41, 41, 21        # :31-34
PUNY = 1.0e-11


def _w1(a):                                       # :626-640
    return (-223.87569446 + 2361.2198663 * a - 10606.56079975 * a * a + 26315.50025642 * a * a * a - 38948.30444297 * a * a * a * a
            + 34397.72407466 * a * a * a * a * a - 16789.98003081 * a * a * a * a * a * a + 3495.82839237 * a * a * a * a * a * a * a)


def _w2(a):                                       # :645-659
    return (-6670.68911883 + 70222.33061536 * a - 314871.71525448 * a * a + 779570.02793492 * a * a * a - 1151098.82436864 * a * a * a * a
            + 1013896.59464498 * a * a * a * a * a - 493379.44906738 * a * a * a * a * a * a + 102356.551518 * a * a * a * a * a * a * a)


def _kernels(x, y, z, p):
    """the six integrands (s11kr, s12kr, s22kr, s11ks, s12ks, s22ks) at broadcastable x, y, z"""
    pih = 0.5 * np.pi
    cos, sin, tan = np.cos, np.sin, np.tan
    n1t2 = [cos(z + pih - p) * cos(z + p), cos(z + pih - p) * sin(z + p), sin(z + pih - p) * cos(z + p), sin(z + pih - p) * sin(z + p)]
    n2t1 = [cos(z - pih + p) * cos(z - p), cos(z - pih + p) * sin(z - p), sin(z - pih + p) * cos(z - p), sin(z - pih + p) * sin(z - p)]
    t1t2 = [cos(z - p) * cos(z + p), cos(z - p) * sin(z + p), sin(z - p) * cos(z + p), sin(z - p) * sin(z + p)]
    t2t1 = [cos(z + p) * cos(z - p), cos(z + p) * sin(z - p), sin(z + p) * cos(z - p), sin(z + p) * sin(z - p)]
    d11 = cos(y) * cos(y) * (cos(x) + sin(x) * tan(y) * tan(y))
    d12 = cos(y) * cos(y) * tan(y) * (-cos(x) + sin(x))
    d22 = cos(y) * cos(y) * (sin(x) + cos(x) * tan(y) * tan(y))
    II = lambda t: t[0] * d11 + (t[1] + t[2]) * d12 + t[3] * d22
    IIn1t2, IIn2t1, IIt1t2 = II(n1t2), II(n2t1), II(t1t2)
    He1 = np.where(-IIn1t2 >= PUNY, 1.0, 0.0)
    He2 = np.where(-IIn2t1 >= PUNY, 1.0, 0.0)
    sg = np.where(IIt1t2 + PUNY >= 0.0, 1.0, -1.0)          # sign(c1, IIt1t2 + puny)
    s11r = -He1 * n1t2[0] - He2 * n2t1[0]
    s12r = 0.5 * ((-He1 * n1t2[1] - He2 * n2t1[1]) + (-He1 * n1t2[2] - He2 * n2t1[2]))
    s22r = -He1 * n1t2[3] - He2 * n2t1[3]
    s11s = sg * (He1 * t1t2[0] + He2 * t2t1[0])
    s12s = 0.5 * (sg * (He1 * t1t2[1] + He2 * t2t1[1]) + sg * (He1 * t1t2[2] + He2 * t2t1[2]))
    s22s = sg * (He1 * t1t2[3] + He2 * t2t1[3])
    return [s11r, s12r, s22r, s11s, s12s, s22s]


def eap_tables(nx_yield=NX_YIELD, ny_yield=NY_YIELD, na_yield=NA_YIELD, nz=100):
    """(s11r, s12r, s22r, s11s, s12s, s22s), each float64 [na_yield][ny_yield][nx_yield]"""
    pi, eps6 = np.pi, 1.0e-6
    pih, piq, phi = 0.5 * pi, 0.25 * pi, pi / 12.0
    da = 0.5 / (na_yield - 1); ainit = 0.5 - da
    dx = pi / (nx_yield - 1); xinit = pi + piq - dx
    dz = pi / nz; zinit = -pih
    dy = pi / (ny_yield - 1); yinit = -dy
    x = (xinit + np.arange(1, nx_yield + 1) * dx)[None, None, :]
    y = (yinit + np.arange(1, ny_yield + 1) * dy)[None, :, None]
    out = [np.zeros((na_yield, ny_yield, nx_yield)) for _ in range(6)]
    a = (ainit + np.arange(1, na_yield) * da)[:, None, None]          # ia = 1 .. na_yield-1
    for iz in range(1, nz + 1):                                       # the sum over z in the reference's order
        z = zinit + iz * dz
        K = _kernels(x, y, z, phi)
        wgt = 1 * _w1(a) * np.exp(-_w2(a) * z * z)
        for t in range(6):
            out[t][:na_yield - 1] += wgt * K[t] * dz / np.sin(2.0 * phi)
    K = _kernels(x, y, 0.0, phi)                                      # ia = na_yield (:604-609)
    for t in range(6):
        out[t][na_yield - 1] = 0.5 * K[t][0] / np.sin(2.0 * phi)
        out[t][np.abs(out[t]) < eps6] = 0.0
    return [np.ascontiguousarray(o) for o in out]
