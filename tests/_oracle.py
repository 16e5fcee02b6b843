"""ctypes bindings for the CPU oracle (oracle/libdfta_oracle.so) and, when present, the compiled
reference (oracle/_ref/libdfta_ref.so).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this module; the product package never does."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# DFTA_ORACLE_OMP=1 (bench.py's level-parallel CPU leg, tests/test_oracle_golden.py): the same source built with -fopenmp
ORACLE_OMP = os.environ.get("DFTA_ORACLE_OMP") == "1"
ORACLE_SO = os.path.join(ORACLE_DIR, "libdfta_oracle_omp.so" if ORACLE_OMP else "libdfta_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdfta_ref.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_lp = C.POINTER(C.c_long)


def dp(a):
    return a.ctypes.data_as(c_dp)


def ip(a):
    return a.ctypes.data_as(c_ip)


class Grid(C.Structure):
    _fields_ = [("N", C.c_int), ("delta", C.c_double), ("Rmax", C.c_double), ("Rp", C.c_double),
                ("twodelta", C.c_double), ("Rp2delta2", C.c_double), ("delta2p4", C.c_double)]


class UGrid(C.Structure):
    _fields_ = [("N", C.c_int), ("Rmax", C.c_double), ("h", C.c_double)]


class Level(C.Structure):
    _fields_ = [("n", C.c_int), ("l", C.c_int), ("occ", C.c_int), ("E", C.c_double),
                ("top", C.c_double), ("bottom", C.c_double), ("n_count", C.c_int), ("n_zero", C.c_int),
                ("converged", C.c_int), ("matchPoint", C.c_long)]


class Energies(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("Etotal", "Ekinetic", "Ecoul", "Enuclear", "Exc",
                                          "Eelectronic", "Ehartree", "eExcDif", "Epotential")]


class Poisson(C.Structure):
    _fields_ = [("levels", C.c_int), ("deltaGrid", C.c_double), ("n", c_ip),
                ("Phi", C.POINTER(c_dp)), ("Src", C.POINTER(c_dp)), ("dlev", c_dp),
                ("lowB", C.c_double), ("highB", C.c_double),
                ("n_gs", C.c_long), ("n_restrict", C.c_long), ("n_prolong", C.c_long), ("n_vcycles", C.c_long)]


class Scf(C.Structure):
    _fields_ = [("lsda", C.c_int), ("Z", C.c_int), ("mgLevels", C.c_int),
                ("alpha", C.c_double), ("MaxR", C.c_double), ("deltaGrid", C.c_double),
                ("g", Grid), ("ps", C.POINTER(Poisson)), ("nla", C.c_int), ("nlb", C.c_int),
                ("la", Level * 32), ("lb", Level * 32),
                ("density", c_dp), ("densityA", c_dp), ("densityB", c_dp), ("potA", c_dp), ("potB", c_dp),
                ("U", c_dp), ("Vexc", c_dp), ("va", c_dp), ("vb", c_dp), ("eexc", c_dp), ("newDensity", c_dp),
                ("tmp", c_dp * 4), ("Eold", C.c_double), ("lastTimeConverged", C.c_int), ("step", C.c_int),
                ("finished", C.c_int), ("chained", C.c_int)]


def build_oracle(force=False):
    """Compile oracle/libdfta_oracle.so (gcc) and, if /root/reference exists, oracle/_ref."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "dfta_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, os.path.basename(ORACLE_SO)], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/DFTAtom"):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    G = C.POINTER(Grid)
    sig = {
        "dfo_num_nodes": (C.c_int, [C.c_int]),
        "dfo_set_level_threads": (None, [C.c_int]),
        "dfo_get_level_threads": (C.c_int, []),
        "dfo_grid_init": (None, [G, C.c_int, C.c_double, C.c_double]),
        "dfo_position": (C.c_double, [G, C.c_long]),
        "dfo_tables_enable": (None, [G]),
        "dfo_tables_disable": (None, []),
        "dfo_veff": (C.c_double, [G, c_dp, C.c_uint, C.c_long]),
        "dfo_f": (C.c_double, [G, c_dp, C.c_uint, C.c_double, C.c_long]),
        "dfo_far": (C.c_double, [G, C.c_double, C.c_double]),
        "dfo_zero": (C.c_double, [G, C.c_double, C.c_uint]),
        "dfo_max_radius_index": (C.c_long, [G, C.c_double, C.c_long]),
        "dfo_count_nodes": (C.c_int, [G, c_dp, C.c_uint, C.c_double, C.c_long, c_lp, c_lp]),
        "dfo_solution_in_zero": (C.c_double, [G, c_dp, C.c_uint, C.c_double, c_lp]),
        "dfo_u0_yardstick": (C.c_double, [G, c_dp, C.c_uint, C.c_double, C.c_long, C.c_int]),
        "dfo_match": (C.c_long, [G, c_dp, C.c_uint, C.c_double, c_dp, c_lp]),
        "dfo_locate_interval": (None, [G, c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_double, c_ip]),
        "dfo_normalize_nonuniform": (None, [G, c_dp]),
        "dfo_loop_over_levels": (C.c_int, [G, c_dp, C.POINTER(Level), C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp]),
        "dfo_calculate_density": (C.c_int, [G, c_dp, C.POINTER(Level), C.c_int, c_dp, C.c_double, c_dp, c_dp,
                                            C.c_double, C.c_int, c_dp]),
        "dfo_poisson_create": (C.POINTER(Poisson), [C.c_int, C.c_double]),
        "dfo_poisson_destroy": (None, [C.POINTER(Poisson)]),
        "dfo_gauss_seidel": (C.c_double, [C.POINTER(Poisson), C.c_int]),
        "dfo_iterate_gs": (C.c_double, [C.POINTER(Poisson), C.c_int, C.c_double, C.c_int]),
        "dfo_restrict": (None, [C.POINTER(Poisson), C.c_int]),
        "dfo_prolong": (None, [c_dp, C.c_int, c_dp]),
        "dfo_initialize": (None, [C.POINTER(Poisson), C.c_double]),
        "dfo_vcycle": (C.c_double, [C.POINTER(Poisson), C.c_double, C.c_int]),
        "dfo_full_cycle": (C.c_double, [C.POINTER(Poisson), C.c_double, C.c_double]),
        "dfo_solve_poisson_nonuniform": (C.c_double, [C.POINTER(Poisson), C.c_int, C.c_double, c_dp, c_dp]),
        "dfo_vwn_vexc": (None, [c_dp, c_dp, C.c_size_t]),
        "dfo_vwn_eexcdif": (None, [c_dp, c_dp, C.c_size_t]),
        "dfo_vwn_vexc_lsda": (None, [c_dp, c_dp, c_dp, c_dp, c_dp, C.c_size_t]),
        "dfo_vwn_eexcdif_lsda": (None, [c_dp, c_dp, c_dp, C.c_size_t]),
        "dfo_trapezoid": (C.c_double, [C.c_double, c_dp, C.c_int]),
        "dfo_simpson13": (C.c_double, [C.c_double, c_dp, C.c_int]),
        "dfo_simpson38": (C.c_double, [C.c_double, c_dp, C.c_int]),
        "dfo_boole": (C.c_double, [C.c_double, c_dp, C.c_int]),
        "dfo_romberg": (C.c_double, [C.c_double, c_dp, C.c_int, C.c_double, C.c_int]),
        "dfo_get_subshells": (C.c_int, [C.c_int, C.POINTER(Level)]),
        "dfo_initialize_levels": (None, [C.c_int, c_ip, c_ip, C.POINTER(Level), c_ip, C.POINTER(Level), c_ip]),
        "dfo_ucount_nodes": (C.c_int, [C.POINTER(UGrid), c_dp, C.c_uint, C.c_double, C.c_long, c_lp]),
        "dfo_usolution_in_zero": (C.c_double, [C.POINTER(UGrid), c_dp, C.c_uint, C.c_double]),
        "dfo_umatch": (C.c_long, [C.POINTER(UGrid), c_dp, C.c_uint, C.c_double, c_dp]),
        "dfo_normalize_uniform": (None, [c_dp, C.c_int, C.c_double]),
        "dfo_uloop_over_levels": (C.c_int, [C.POINTER(UGrid), c_dp, C.POINTER(Level), C.c_int, c_dp, c_dp, c_dp]),
        "dfo_solve_poisson_uniform": (C.c_double, [C.POINTER(Poisson), C.c_int, C.c_double, c_dp, c_dp]),
        "dfo_scf_create": (C.POINTER(Scf), [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]),
        "dfo_scf_destroy": (None, [C.POINTER(Scf)]),
        "dfo_scf_step": (C.c_int, [C.POINTER(Scf), C.POINTER(Energies)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _oracle = L
    return L


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is not None:
        return _ref
    L = C.CDLL(REF_SO)
    vp = C.c_void_p
    sig = {
        "ref_numerov_create": (vp, [c_dp, C.c_int, C.c_double, C.c_double]),
        "ref_numerov_destroy": (None, [vp]),
        "ref_count_nodes": (C.c_int, [vp, C.c_uint, C.c_double, C.c_long]),
        "ref_solution_in_zero": (C.c_double, [vp, C.c_uint, C.c_double]),
        "ref_match": (C.c_long, [vp, C.c_uint, C.c_double, c_dp]),
        "ref_max_radius_index": (C.c_long, [vp, C.c_double]),
        "ref_far": (C.c_double, [vp, C.c_double, C.c_double]),
        "ref_zero": (C.c_double, [vp, C.c_double, C.c_uint]),
        "ref_f": (C.c_double, [vp, C.c_uint, C.c_double, C.c_long]),
        "ref_rp": (C.c_double, [vp]),
        "ref_loop_over_levels": (C.c_int, [vp, C.c_int, c_ip, c_ip, c_ip, c_dp, c_dp, c_dp, c_dp, C.c_double]),
        "ref_locate_interval": (None, [vp, c_dp, c_dp, C.c_int, C.c_int]),
        "ref_normalize_nonuniform": (None, [c_dp, C.c_int, C.c_double, C.c_double]),
        "ref_poisson_create": (vp, [C.c_int, C.c_double]),
        "ref_poisson_destroy": (None, [vp]),
        "ref_poisson_level_size": (C.c_int, [vp, C.c_int]),
        "ref_poisson_set_level": (None, [vp, C.c_int, c_dp, c_dp]),
        "ref_poisson_get_level": (None, [vp, C.c_int, c_dp, c_dp]),
        "ref_gauss_seidel": (C.c_double, [vp, C.c_int]),
        "ref_restrict": (None, [vp, C.c_int]),
        "ref_prolong": (None, [vp, C.c_int]),
        "ref_poisson_set_boundaries": (None, [vp, C.c_double, C.c_double]),
        "ref_poisson_initialize": (None, [vp, C.c_double]),
        "ref_vcycle": (C.c_double, [vp, C.c_double, C.c_int]),
        "ref_full_cycle": (C.c_double, [vp, C.c_double, C.c_double]),
        "ref_solve_poisson_nonuniform": (None, [vp, C.c_int, C.c_double, c_dp, C.c_int, c_dp]),
        "ref_num_nodes": (C.c_int, [C.c_int]),
        "ref_vwn_vexc": (None, [c_dp, c_dp, C.c_int]),
        "ref_vwn_eexcdif": (None, [c_dp, c_dp, C.c_int]),
        "ref_vwn_vexc_lsda": (None, [c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int]),
        "ref_vwn_eexcdif_lsda": (None, [c_dp, c_dp, c_dp, C.c_int]),
        "ref_integral": (C.c_double, [C.c_int, C.c_double, c_dp, C.c_int]),
        "ref_get_subshells": (C.c_int, [C.c_int, c_ip, c_ip, c_ip]),
        "ref_initialize_levels": (None, [C.c_int, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip]),
        "ref_calculate": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_long]),
        "ref_calculate_hp": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_long]),
        "ref_calculate_hp_steps": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                              C.c_char_p, C.c_long]),
        "ref_chachiyo": (None, [C.c_int, c_dp, c_dp, c_dp, C.c_int]),
        "ref_unumerov_create": (vp, [c_dp, C.c_int, C.c_double]),
        "ref_unumerov_destroy": (None, [vp]),
        "ref_ucount_nodes": (C.c_int, [vp, C.c_uint, C.c_double, C.c_long]),
        "ref_usolution_in_zero": (C.c_double, [vp, C.c_uint, C.c_double]),
        "ref_umatch": (C.c_long, [vp, C.c_uint, C.c_double, c_dp]),
        "ref_uloop_over_levels": (C.c_int, [vp, C.c_int, c_ip, c_ip, c_ip, c_dp, c_dp, c_dp, c_dp]),
        "ref_normalize_uniform": (None, [c_dp, C.c_int, C.c_double]),
        "ref_solve_poisson_uniform": (None, [vp, C.c_int, C.c_double, c_dp, C.c_int, c_dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _ref = L
    return L


# ---------------------------------------------------------------------------------------------
# convenience wrappers (numpy in / numpy out) used by tests, golden generation and cpu_baseline
# ---------------------------------------------------------------------------------------------

def make_grid(levels, delta, Rmax):
    g = Grid()
    o = oracle()
    o.dfo_grid_init(C.byref(g), o.dfo_num_nodes(levels), delta, Rmax)
    return g


def make_ugrid(levels, Rmax):
    N = oracle().dfo_num_nodes(levels)
    return UGrid(N, Rmax, Rmax / (N - 1))


def grid_r(g):
    """r_i = Rp (exp(i delta) - 1) with the reference's expression order."""
    o = oracle()
    return np.array([o.dfo_position(C.byref(g), i) for i in range(g.N)])


def coulomb_potential(g, Z):
    r = grid_r(g)
    V = np.zeros(g.N)
    V[1:] = -Z / r[1:]
    return V


def levels_array(items):
    arr = (Level * len(items))()
    for k, (n, l, occ) in enumerate(items):
        arr[k].n, arr[k].l, arr[k].occ = n, l, occ
    return arr


def subshells(Z):
    o = oracle()
    arr = (Level * 32)()
    cnt = o.dfo_get_subshells(Z, arr)
    return [(arr[i].n, arr[i].l, arr[i].occ) for i in range(cnt)]
