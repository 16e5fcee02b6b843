"""dftatom_amd -- MI355X (gfx950) radial-DFT inner loop behind a C ABI.

This package is a thin ctypes binding over ``dftatom_amd/libdftatom_hip.so`` (built from
``dftatom_amd/csrc`` by ``__graft_entry__.build()`` / ``make -C dftatom_amd/csrc``).  There is no CPU
fallback: if the shared library is missing, or no HIP device is usable, every entry point raises.

When PyTorch is used in the same process (bench.py: device memory, streams, torch.distributed), import
torch BEFORE this module so that both share one HIP runtime (same libamdhip64 soname).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFTA_LIB_PATH") or os.path.join(_HERE, "libdftatom_hip.so")      # DFTA_LIB_PATH: measurement builds

OK = 0
SWEEP_COUNT, SWEEP_ZERO = 0, 1
SWEEP_KERNEL_AUTO, SWEEP_KERNEL_FUSED, SWEEP_KERNEL_PIPELINED = 0, 1, 2
BOUNDARY_DEVICE, BOUNDARY_HOST = 0, 1
LEVELS_CHAINED, LEVELS_BATCHED = 0, 1
LEVELS_SCAN_SWEEPS = 0x10          # OR-ed into the mode of solve_levels: tolerance mode of the sweeps (transfer-matrix scan)
SWEEPS_EXACT, SWEEPS_TOLERANCE = 0, 1
ABI_VERSION = 6
LEVEL_CONVERGED, LEVEL_ITERATION_CAP, LEVEL_FIXED_POINT, LEVEL_U0_NONFINITE = 1, 2, 4, 8
POISSON_DEFAULT, POISSON_EXACT, POISSON_TOLERANCE, POISSON_ADAPTIVE = -1, 0, 1, 2    # dfta_poisson_create_ex / dfta_scf_options::poisson_mode
INT_TRAPEZOID, INT_SIMPSON13, INT_SIMPSON38, INT_BOOLE, INT_ROMBERG = range(5)
XC_VWN, XC_CHACHIYO, XC_CHACHIYO_IMPROVED = range(3)
AUFBAU_REFERENCE, AUFBAU_TRANSITION_METALS = range(2)
RECORD_DOUBLES = 64

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_lp = C.POINTER(C.c_long)
vp = C.c_void_p


class DftaError(RuntimeError):
    pass


class LevelResult(C.Structure):
    _fields_ = [("E", C.c_double), ("top", C.c_double), ("bottom", C.c_double), ("n_count", C.c_int),
                ("n_zero", C.c_int), ("converged", C.c_int), ("matchPoint", C.c_int), ("status", C.c_int)]


class Energies(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("Etotal", "Ekinetic", "Ecoul", "Enuclear", "Exc",
                                          "Eelectronic", "Ehartree", "eExcDif", "Epotential")]

    def as_list(self):
        return [self.Etotal, self.Ekinetic, self.Ecoul, self.Enuclear, self.Exc]


class ScfOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("integrator", C.c_int), ("functional", C.c_int), ("aufbau", C.c_int), ("poisson_mode", C.c_int), ("sweep_mode", C.c_int)]


class StepStats(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("levels_fallbacks", C.c_int), ("sweeps_issued", C.c_long), ("sweeps_reference", C.c_long), ("points_traversed", C.c_long),
                ("vcycles", C.c_long), ("rounds", C.c_int), ("ms_levels", C.c_float), ("ms_poisson", C.c_float),
                ("ms_tail", C.c_float), ("ms_sweep_kernels", C.c_float), ("ms_poisson_kernel", C.c_float),
                ("sweeps_reference_executed", C.c_long), ("points_reference", C.c_long),
                ("levels_layout", C.c_int), ("poisson_groups", C.c_int)]


# every symbol include/dftatom_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "dfta_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "dfta_ctx_destroy": (None, [vp]),
    "dfta_ctx_synchronize": (C.c_int, [vp]),
    "dfta_last_error": (C.c_char_p, [vp]),
    "dfta_version": (C.c_char_p, []),
    "dfta_abi_version": (C.c_int, []),
    "dfta_ctx_device_info": (C.c_int, [vp, c_ip, C.c_char_p, C.c_int]),
    "dfta_ctx_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "dfta_ctx_set_sweep_kernel": (C.c_int, [vp, C.c_int]),
    "dfta_grid_create": (C.c_int, [vp, C.c_int, C.c_double, C.c_double, C.POINTER(vp)]),
    "dfta_grid_create_uniform": (C.c_int, [vp, C.c_int, C.c_double, C.POINTER(vp)]),
    "dfta_grid_is_uniform": (C.c_int, [vp]),
    "dfta_grid_destroy": (None, [vp]),
    "dfta_grid_num_nodes": (C.c_int, [vp]),
    "dfta_grid_rp": (C.c_double, [vp]),
    "dfta_grid_get_r": (C.c_int, [vp, c_dp]),
    "dfta_num_nodes": (C.c_int, [C.c_int]),
    "dfta_numerov_sweeps": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, c_dp, C.c_int, c_ip, c_ip, c_dp, c_ip,
                                      c_ip, c_dp, c_ip, c_ip]),
    "dfta_numerov_sweeps_scan": (C.c_int, [vp, vp, C.c_int, C.c_int, c_dp, C.c_int, c_ip, c_ip, c_dp, c_ip, c_ip, c_dp, c_ip, c_ip, c_ip]),
    "dfta_numerov_sweeps_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, c_ip, c_ip, c_ip, vp, vp, vp, vp, vp,
                                          vp, vp, vp, vp]),
    "dfta_numerov_match": (C.c_int, [vp, vp, C.c_int, C.c_int, c_dp, C.c_int, c_ip, c_ip, c_dp, c_dp, c_lp]),
    "dfta_potential_create": (C.c_int, [vp, vp, c_dp, C.POINTER(vp)]),
    "dfta_potential_update": (C.c_int, [vp, c_dp]),
    "dfta_potential_destroy": (None, [vp]),
    "dfta_potential_sweeps": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_ip, c_dp, c_ip, c_ip, c_dp, c_ip, c_ip]),
    "dfta_potential_match": (C.c_int, [vp, C.c_int, c_ip, c_dp, c_dp, c_lp]),
    "dfta_solve_levels": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_int, c_ip, c_ip, c_ip, c_ip,
                                    C.POINTER(LevelResult), c_dp, c_dp, c_dp, c_lp]),
    "dfta_poisson_create": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp)]),
    "dfta_poisson_destroy": (None, [vp]),
    "dfta_poisson_solve": (C.c_int, [vp, c_ip, c_dp, c_dp, c_ip, c_dp]),
    "dfta_poisson_solve_dev": (C.c_int, [vp, vp, vp, vp]),
    "dfta_poisson_group_info": (C.c_int, [vp, c_ip, c_ip, c_ip]),
    "dfta_scf_poisson_info": (C.c_int, [vp, c_ip, c_ip, c_ip]),
    "dfta_poisson_level_size": (C.c_int, [vp, C.c_int]),
    "dfta_poisson_set_level": (C.c_int, [vp, C.c_int, c_dp, c_dp]),
    "dfta_poisson_get_level": (C.c_int, [vp, C.c_int, c_dp, c_dp]),
    "dfta_poisson_gauss_seidel": (C.c_int, [vp, C.c_int, C.c_int, c_dp]),
    "dfta_poisson_iterate_gs": (C.c_int, [vp, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "dfta_poisson_restrict": (C.c_int, [vp, C.c_int]),
    "dfta_poisson_prolong": (C.c_int, [vp, C.c_int]),
    "dfta_poisson_vcycle": (C.c_int, [vp, c_dp]),
    "dfta_poisson_full_cycle": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, C.c_double, c_dp, c_ip]),
    "dfta_scf_set_integrator": (C.c_int, [vp, C.c_int]),
    "dfta_vwn_lda": (C.c_int, [vp, c_dp, C.c_size_t, c_dp, c_dp]),
    "dfta_vwn_lsda": (C.c_int, [vp, c_dp, c_dp, C.c_size_t, c_dp, c_dp, c_dp, c_dp]),
    "dfta_integrate": (C.c_int, [vp, C.c_int, C.c_double, c_dp, C.c_int, C.POINTER(C.c_double)]),
    "dfta_scf_create": (C.c_int, [vp, vp, C.c_int, C.c_int, c_ip, C.c_double, C.c_int, C.c_int, C.POINTER(vp)]),
    "dfta_scf_create_ex": (C.c_int, [vp, vp, C.c_int, C.c_int, c_ip, C.c_double, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "dfta_scf_destroy": (None, [vp]),
    "dfta_scf_step": (C.c_int, [vp, C.POINTER(StepStats)]),
    "dfta_scf_get_energies": (C.c_int, [vp, C.POINTER(Energies), c_ip]),
    "dfta_scf_info": (C.c_int, [vp, c_ip, c_ip, c_lp]),
    "dfta_scf_num_levels": (C.c_int, [vp, C.c_int, C.c_int]),
    "dfta_scf_get_levels": (C.c_int, [vp, C.c_int, C.c_int, c_ip, c_ip, c_ip, c_dp, c_ip]),
    "dfta_scf_get_level_status": (C.c_int, [vp, C.c_int, C.c_int, c_ip, c_ip, c_ip]),
    "dfta_scf_get_array": (C.c_int, [vp, C.c_int, C.c_int, c_dp]),
    "dfta_scf_get_records_dev": (C.c_int, [vp, vp]),
    "dfta_get_subshells": (C.c_int, [C.c_int, c_ip, c_ip, c_ip, C.c_int]),
    "dfta_get_subshells_ex": (C.c_int, [C.c_int, C.c_int, c_ip, c_ip, c_ip, C.c_int]),
    "dfta_split_spin_ex": (C.c_int, [C.c_int, C.c_int, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, C.c_int]),
    "dfta_chachiyo_lda": (C.c_int, [vp, C.c_int, c_dp, C.c_size_t, c_dp, c_dp]),
    "dfta_split_spin": (C.c_int, [C.c_int, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, C.c_int]),
    "dfta_ctx_measure_hbm": (C.c_int, [vp, C.c_size_t, C.c_int, c_dp, c_dp]),
    "dfta_poisson_create_ex": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "dfta_poisson_mode": (C.c_int, [vp]),
}

_lib = None


def load():
    """Load libdftatom_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DftaError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        f.restype = res
        f.argtypes = args
    if lib.dfta_abi_version() != ABI_VERSION:      # the struct mirrors above follow the header (struct_size first, growth at the end)
        raise DftaError("%s was built from another version of include/dftatom_hip.h (ABI %d, binding %d): rebuild it"
                        % (LIB_PATH, lib.dfta_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _ip(a):
    return a.ctypes.data_as(c_ip)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """Device context (dfta_ctx).  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream; None -> stream owned by the context."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = vp()
        rc = self.lib.dfta_ctx_create(device, vp(stream) if stream else None, C.byref(h))
        if rc != OK:
            raise DftaError("dfta_ctx_create failed with status %d (no usable HIP device?)" % rc)
        self.h = h

    def check(self, rc):
        if rc != OK:
            raise DftaError("status %d: %s" % (rc, self.lib.dfta_last_error(self.h).decode()))

    def synchronize(self):
        self.check(self.lib.dfta_ctx_synchronize(self.h))

    def device_info(self):
        ncu = C.c_int()
        name = C.create_string_buffer(128)
        self.check(self.lib.dfta_ctx_device_info(self.h, C.byref(ncu), name, 128))
        return ncu.value, name.value.decode()

    def set_sweep_kernel(self, which):
        """SWEEP_KERNEL_AUTO / _FUSED / _PIPELINED: which (bit-identical) Numerov sweep kernel later calls launch."""
        self.check(self.lib.dfta_ctx_set_sweep_kernel(self.h, int(which)))

    def measure_hbm(self, doubles_per_array=1 << 27, reps=5):
        """attainable HBM bandwidth (GB/s) of the device: (copy, triad) -- measurement aid for the roofline denominators"""
        c, t = C.c_double(), C.c_double()
        self.check(self.lib.dfta_ctx_measure_hbm(self.h, int(doubles_per_array), int(reps), C.byref(c), C.byref(t)))
        return c.value, t.value

    def last_kernel_ms(self):
        ms = C.c_float()
        self.check(self.lib.dfta_ctx_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.lib.dfta_ctx_destroy(self.h)
            self.h = None


class Grid:
    """Logarithmic grid r_i = Rp (exp(i delta) - 1), or -- delta = None / 0 -- the uniform grid r_i = i Rmax / (N - 1)."""

    def __init__(self, ctx, mg_levels, delta, Rmax):
        self.ctx = ctx
        h = vp()
        self.uniform = not delta
        if self.uniform:
            ctx.check(ctx.lib.dfta_grid_create_uniform(ctx.h, mg_levels, Rmax, C.byref(h)))
        else:
            ctx.check(ctx.lib.dfta_grid_create(ctx.h, mg_levels, delta, Rmax, C.byref(h)))
        self.h = h
        self.levels, self.delta, self.Rmax = mg_levels, (delta or 0.0), Rmax
        self.N = ctx.lib.dfta_grid_num_nodes(h)
        self.Rp = ctx.lib.dfta_grid_rp(h)

    def r(self):
        out = np.zeros(self.N)
        self.ctx.check(self.ctx.lib.dfta_grid_get_r(self.h, _dp(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.lib.dfta_grid_destroy(self.h)
            self.h = None


def numerov_sweeps(ctx, grid, kind, V, l, E, limit=None, vidx=None, boundary=BOUNDARY_HOST):
    """Batched SolveSchrodingerCountNodes / SolutionInZero.  Returns dict(count, u0, start, trip)."""
    V = _f64(V).reshape(-1, grid.N)
    l = _i32(l)
    E = _f64(E)
    nt = len(E)
    lim = _i32(limit) if limit is not None else np.zeros(nt, np.int32)
    vi = _i32(vidx) if vidx is not None else np.zeros(nt, np.int32)
    count = np.zeros(nt, np.int32)
    u0 = np.zeros(nt)
    start = np.zeros(nt, np.int32)
    trip = np.zeros(nt, np.int32)
    ctx.check(ctx.lib.dfta_numerov_sweeps(ctx.h, grid.h, kind, boundary, V.shape[0], _dp(V), nt, _ip(vi), _ip(l), _dp(E),
                                          _ip(lim), _ip(count), _dp(u0), _ip(start), _ip(trip)))
    return {"count": count, "u0": u0, "start": start, "trip": trip}


def numerov_sweeps_scan(ctx, grid, kind, V, l, E, limit=None, vidx=None):
    """Tolerance mode of the same sweeps (transfer-matrix scan, one workgroup per trial).  Returns dict(count, u0, start, trip, fallback)."""
    V = _f64(V).reshape(-1, grid.N)
    l = _i32(l)
    E = _f64(E)
    nt = len(E)
    lim = _i32(limit) if limit is not None else np.zeros(nt, np.int32)
    vi = _i32(vidx) if vidx is not None else np.zeros(nt, np.int32)
    count, start, trip, fb = (np.zeros(nt, np.int32) for _ in range(4))
    u0 = np.zeros(nt)
    ctx.check(ctx.lib.dfta_numerov_sweeps_scan(ctx.h, grid.h, kind, V.shape[0], _dp(V), nt, _ip(vi), _ip(l), _dp(E), _ip(lim), _ip(count),
                                               _dp(u0), _ip(start), _ip(trip), _ip(fb)))
    return {"count": count, "u0": u0, "start": start, "trip": trip, "fallback": fb}


class Potential:
    """A potential resident on the device with its slot tables (dfta_potential): per-call sweeps without the upload and table build."""

    def __init__(self, ctx, grid, V):
        self.ctx, self.grid = ctx, grid
        h = vp()
        V = _f64(V)
        ctx.check(ctx.lib.dfta_potential_create(ctx.h, grid.h, _dp(V), C.byref(h)))
        self.h = h

    def update(self, V):
        V = _f64(V)
        self.ctx.check(self.ctx.lib.dfta_potential_update(self.h, _dp(V)))

    def sweeps(self, kind, l, E, limit=None, sweep_mode=SWEEPS_EXACT):
        l, E = _i32(l), _f64(E)
        nt = len(E)
        lim = _i32(limit) if limit is not None else np.zeros(nt, np.int32)
        count, start, trip = (np.zeros(nt, np.int32) for _ in range(3))
        u0 = np.zeros(nt)
        self.ctx.check(self.ctx.lib.dfta_potential_sweeps(self.h, kind, sweep_mode, nt, _ip(l), _dp(E), _ip(lim), _ip(count), _dp(u0), _ip(start), _ip(trip)))
        return {"count": count, "u0": u0, "start": start, "trip": trip}

    def match(self, l, E):
        l, E = _i32(l), _f64(E)
        nt = len(E)
        psi = np.zeros((nt, self.grid.N))
        mp = np.zeros(nt, np.int64)
        self.ctx.check(self.ctx.lib.dfta_potential_match(self.h, nt, _ip(l), _dp(E), _dp(psi), mp.ctypes.data_as(c_lp)))
        return psi, mp

    def close(self):
        if self.h:
            self.ctx.lib.dfta_potential_destroy(self.h)
            self.h = None


def numerov_match(ctx, grid, V, l, E, vidx=None, boundary=BOUNDARY_HOST):
    V = _f64(V).reshape(-1, grid.N)
    l = _i32(l)
    E = _f64(E)
    nt = len(E)
    vi = _i32(vidx) if vidx is not None else np.zeros(nt, np.int32)
    psi = np.zeros((nt, grid.N))
    mp = np.zeros(nt, np.int64)
    ctx.check(ctx.lib.dfta_numerov_match(ctx.h, grid.h, boundary, V.shape[0], _dp(V), nt, _ip(vi), _ip(l), _dp(E),
                                         _dp(psi), mp.ctypes.data_as(c_lp)))
    return psi, mp


def get_subshells(Z, aufbau=AUFBAU_REFERENCE):
    lib = load()
    n, l, occ = (np.zeros(32, np.int32) for _ in range(3))
    c = lib.dfta_get_subshells_ex(Z, aufbau, _ip(n), _ip(l), _ip(occ), 32)
    if c < 0:
        raise DftaError("dfta_get_subshells(%d) failed" % Z)
    return [(int(n[i]), int(l[i]), int(occ[i])) for i in range(c)]


def solve_levels(ctx, grid, V, levels, bottom0, vidx=None, mode=LEVELS_BATCHED, tree_depth=0, want_psi=False, hints=None):
    """Device-side LoopOverLevels for `levels` = [(n, l, occ), ...] on potentials V (nV x N).
    Returns dict(E, top, bottom, n_count, n_zero, converged, matchPoint, newDensity, Eelectronic, psi, issued)."""
    V = _f64(V).reshape(-1, grid.N)
    nV = V.shape[0]
    nl = len(levels)
    n = _i32([a for a, _, _ in levels])
    l = _i32([b for _, b, _ in levels])
    occ = _i32([c for _, _, c in levels])
    vi = _i32(vidx) if vidx is not None else np.zeros(nl, np.int32)
    b0 = _f64(np.broadcast_to(np.asarray(bottom0, dtype=np.float64), (nV,)))
    res = (LevelResult * nl)()
    nd = np.zeros((nV, grid.N))
    eel = np.zeros(nV)
    psi = np.zeros((nl, grid.N)) if want_psi else None
    issued = C.c_long(0)
    hh = _f64(hints) if hints is not None else None
    ctx.check(ctx.lib.dfta_solve_levels(ctx.h, grid.h, mode, tree_depth, nV, _dp(V), _dp(b0), _dp(hh) if hh is not None else None,
                                        nl, _ip(vi), _ip(n), _ip(l),
                                        _ip(occ), res, _dp(nd), _dp(eel), _dp(psi) if want_psi else None, C.byref(issued)))
    out = {k: np.array([getattr(res[i], k) for i in range(nl)]) for k in
           ("E", "top", "bottom", "n_count", "n_zero", "converged", "matchPoint", "status")}
    out.update(newDensity=nd, Eelectronic=eel, psi=psi, issued=issued.value)
    return out


def integrate(ctx, rule, delta, values):
    v = _f64(values)
    out = C.c_double()
    ctx.check(ctx.lib.dfta_integrate(ctx.h, rule, delta, _dp(v), len(v), C.byref(out)))
    return out.value


class Poisson:
    """DFT::PoissonSolver for a batch of atoms on one grid (dfta_poisson)."""

    def __init__(self, ctx, grid, batch=1, mode=POISSON_DEFAULT):
        self.ctx, self.grid, self.batch = ctx, grid, batch
        h = vp()
        if mode == POISSON_DEFAULT:
            ctx.check(ctx.lib.dfta_poisson_create(ctx.h, grid.h, batch, C.byref(h)))
        else:
            ctx.check(ctx.lib.dfta_poisson_create_ex(ctx.h, grid.h, batch, int(mode), C.byref(h)))
        self.h = h
        self.mode = ctx.lib.dfta_poisson_mode(h)

    def solve(self, Z, density):
        """SolvePoissonNonUniform: density (batch x N) -> U (batch x N), vcycles, err."""
        Z = _i32(np.atleast_1d(Z))
        rho = _f64(density).reshape(self.batch, self.grid.N)
        U = np.zeros_like(rho)
        vc = np.zeros(self.batch, np.int32)
        err = np.zeros(self.batch)
        self.ctx.check(self.ctx.lib.dfta_poisson_solve(self.h, _ip(Z), _dp(rho), _dp(U), _ip(vc), _dp(err)))
        return U, vc, err

    def group_info(self):
        """(workgroups per atom, degraded to one workgroup per atom?, solves that had to be repeated)"""
        g, d, a = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.dfta_poisson_group_info(self.h, C.byref(g), C.byref(d), C.byref(a)))
        return g.value, bool(d.value), a.value

    def level_size(self, lvl):
        return self.ctx.lib.dfta_poisson_level_size(self.h, lvl)

    def set_level(self, lvl, phi=None, src=None):
        self.ctx.check(self.ctx.lib.dfta_poisson_set_level(self.h, lvl, _dp(_f64(phi)) if phi is not None else None,
                                                           _dp(_f64(src)) if src is not None else None))

    def get_level(self, lvl):
        n = self.level_size(lvl)
        phi, src = np.zeros(n), np.zeros(n)
        self.ctx.check(self.ctx.lib.dfta_poisson_get_level(self.h, lvl, _dp(phi), _dp(src)))
        return phi, src

    def gauss_seidel(self, lvl, sweeps=1):
        err = np.zeros(sweeps)
        self.ctx.check(self.ctx.lib.dfta_poisson_gauss_seidel(self.h, lvl, sweeps, _dp(err)))
        return err

    def iterate_gs(self, lvl, error_min, iterno):
        """IterateGaussSeidel: returns (err of the last sweep, sweeps executed)."""
        err = C.c_double()
        n = C.c_int()
        self.ctx.check(self.ctx.lib.dfta_poisson_iterate_gs(self.h, lvl, error_min, iterno, C.byref(err), C.byref(n)))
        return err.value, n.value

    def restrict(self, lvl):
        self.ctx.check(self.ctx.lib.dfta_poisson_restrict(self.h, lvl))

    def prolong(self, lvl_src):
        self.ctx.check(self.ctx.lib.dfta_poisson_prolong(self.h, lvl_src))

    def vcycle(self):
        err = np.zeros(1)
        self.ctx.check(self.ctx.lib.dfta_poisson_vcycle(self.h, _dp(err)))
        return err[0]

    def full_cycle(self, low, high, error_min=1e-3, error_min_last=1e-14):
        """SetBoundaries + FullCycle on the source left by the last solve: returns (err, V-cycles)."""
        err, vc = C.c_double(), C.c_int()
        self.ctx.check(self.ctx.lib.dfta_poisson_full_cycle(self.h, low, high, error_min, error_min_last, C.byref(err), C.byref(vc)))
        return err.value, vc.value

    def close(self):
        if self.h:
            self.ctx.lib.dfta_poisson_destroy(self.h)
            self.h = None


def vwn_lda(ctx, n):
    n = _f64(n)
    v, e = np.zeros_like(n), np.zeros_like(n)
    ctx.check(ctx.lib.dfta_vwn_lda(ctx.h, _dp(n), n.size, _dp(v), _dp(e)))
    return v, e


def chachiyo_lda(ctx, n, improved=True):
    n = _f64(n)
    v, e = np.zeros_like(n), np.zeros_like(n)
    ctx.check(ctx.lib.dfta_chachiyo_lda(ctx.h, int(improved), _dp(n), n.size, _dp(v), _dp(e)))
    return v, e


def vwn_lsda(ctx, na, nb):
    na, nb = _f64(na), _f64(nb)
    r, va, vb, e = (np.zeros_like(na) for _ in range(4))
    ctx.check(ctx.lib.dfta_vwn_lsda(ctx.h, _dp(na), _dp(nb), na.size, _dp(r), _dp(va), _dp(vb), _dp(e)))
    return r, va, vb, e


class Scf:
    """Device-resident SCF state of a batch of atoms (dfta_scf): the body of CalculateNonUniformLDA/LSDA."""

    def __init__(self, ctx, grid, Z, lsda=False, alpha=0.5, levels_mode=LEVELS_BATCHED, tree_depth=0, integrator=INT_SIMPSON38,
                 functional=XC_VWN, aufbau=AUFBAU_REFERENCE, poisson_mode=POISSON_DEFAULT, sweep_mode=SWEEPS_EXACT):
        self.ctx, self.grid = ctx, grid
        self.Z = _i32(np.atleast_1d(Z))
        self.natoms = len(self.Z)
        self.lsda = bool(lsda)
        h = vp()
        opt = ScfOptions(C.sizeof(ScfOptions), integrator, functional, aufbau, poisson_mode, sweep_mode)
        ctx.check(ctx.lib.dfta_scf_create_ex(ctx.h, grid.h, int(self.lsda), self.natoms, _ip(self.Z), alpha, levels_mode,
                                             tree_depth, C.cast(C.byref(opt), vp), C.byref(h)))
        self.h = h
        d, nj, tr = C.c_int(), C.c_int(), C.c_long()
        ctx.check(ctx.lib.dfta_scf_info(h, C.byref(d), C.byref(nj), C.byref(tr)))
        self.tree_depth, self.njobs, self.trials_per_round = d.value, nj.value, tr.value

    def step(self, want_stats=True):
        st = StepStats()
        st.struct_size = C.sizeof(StepStats)
        self.ctx.check(self.ctx.lib.dfta_scf_step(self.h, C.byref(st) if want_stats else None))
        return st

    def energies(self):
        e = (Energies * self.natoms)()
        fin = np.zeros(self.natoms, np.int32)
        self.ctx.check(self.ctx.lib.dfta_scf_get_energies(self.h, e, _ip(fin)))
        return [e[i] for i in range(self.natoms)], fin

    def levels(self, atom=0, spin=0):
        cnt = self.ctx.lib.dfta_scf_num_levels(self.h, atom, spin)
        n, l, occ, conv = (np.zeros(max(cnt, 1), np.int32) for _ in range(4))
        E = np.zeros(max(cnt, 1))
        if cnt > 0:
            self.ctx.check(self.ctx.lib.dfta_scf_get_levels(self.h, atom, spin, _ip(n), _ip(l), _ip(occ), _dp(E), _ip(conv)))
        status, nc, nz = (np.zeros(max(cnt, 1), np.int32) for _ in range(3))
        if cnt > 0:
            self.ctx.check(self.ctx.lib.dfta_scf_get_level_status(self.h, atom, spin, _ip(status), _ip(nc), _ip(nz)))
        return {"n": n[:cnt], "l": l[:cnt], "occ": occ[:cnt], "E": E[:cnt], "converged": conv[:cnt], "status": status[:cnt],
                "n_count": nc[:cnt], "n_zero": nz[:cnt]}

    def array(self, which, atom=0):
        out = np.zeros(self.grid.N)
        self.ctx.check(self.ctx.lib.dfta_scf_get_array(self.h, atom, which, _dp(out)))
        return out

    def set_integrator(self, rule):
        """INT_TRAPEZOID .. INT_ROMBERG: quadrature of the energy integrals and of the orbitals' normalisation."""
        self.ctx.check(self.ctx.lib.dfta_scf_set_integrator(self.h, int(rule)))

    def poisson_info(self):
        g, d, a = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.dfta_scf_poisson_info(self.h, C.byref(g), C.byref(d), C.byref(a)))
        return g.value, bool(d.value), a.value

    def records_into(self, device_ptr):
        self.ctx.check(self.ctx.lib.dfta_scf_get_records_dev(self.h, vp(device_ptr)))

    def close(self):
        if self.h:
            self.ctx.lib.dfta_scf_destroy(self.h)
            self.h = None
