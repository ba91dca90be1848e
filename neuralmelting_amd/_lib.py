"""ctypes binding of libnm_hip.so (include/nm.h).  There is no CPU fallback: if the HIP library is
missing or cannot be loaded this module raises, it never substitutes another implementation."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NM_HIP_LIB') or os.path.join(_HERE, 'libnm_hip.so')  # NM_HIP_LIB: dev override (diagnostic builds)

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)

NM_OK, NM_ERR_ARG, NM_ERR_HIP, NM_ERR_STATE, NM_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
NM_EL_LJ, NM_EL_AL = 0, 1
NM_THERMO_COLS, NM_TRACE_COLS, NM_STATS_COLS = 17, 4, 10

# every symbol include/nm.h declares (tests check that the library exports all of them)
SYMBOLS = ('nm_create', 'nm_destroy', 'nm_last_error', 'nm_create_note', 'nm_nslots', 'nm_natoms', 'nm_cus_per_replica', 'nm_heal_count', 'nm_get_const',
           'nm_get_slots', 'nm_set_slots', 'nm_snapshot', 'nm_snapshot_fetch',
           'nm_set_state', 'nm_get_state', 'nm_init_lattice', 'nm_lattice_state', 'nm_set_thermo', 'nm_set_step', 'nm_run_md', 'nm_run_block', 'nm_run_cycles', 'nm_get_thermo', 'nm_adapt',
           'nm_exchange', 'nm_synchronize', 'nm_get_status', 'nm_format_thrm', 'nm_format_traj', 'nm_append_outputs', 'nm_timing_reset', 'nm_timing_get', 'nm_stats_get', 'nm_eval',
           'nm_set_rng_tape', 'nm_set_exchange_tape', 'nm_set_trace', 'nm_get_trace', 'nm_get_perm', 'nm_set_counters',
           'nm_get_exchange_crit')


# include/nm_distr.h
DISTR_SYMBOLS = ('nm_distr_histograms', 'nm_distr_last_error')
# include/nm_parse.h
PARSE_SYMBOLS = ('nm_parse_thrm', 'nm_parse_traj', 'nm_parse_last_error')


class NMConfig(C.Structure):
    _fields_ = [('size', C.c_int32), ('element', C.c_int32), ('natoms', C.c_int32), ('np', C.c_int32),
                ('nt', C.c_int32), ('row0', C.c_int32), ('nrows', C.c_int32), ('nstps', C.c_int32),
                ('bulk', C.c_int32), ('iter_revert', C.c_int32), ('device', C.c_int32), ('seed', C.c_uint32),
                ('ppos', C.c_double), ('pvol', C.c_double), ('P', c_float_p), ('T', c_float_p),
                ('slot0', C.c_int32), ('nslots', C.c_int32)]


_lib = None


def load():
    """load libnm_hip.so and declare the prototypes of include/nm.h"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); this package has no CPU fallback' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.nm_create.argtypes = [C.POINTER(NMConfig), C.POINTER(vp)]
    L.nm_destroy.argtypes = [vp]
    L.nm_last_error.restype = C.c_char_p
    L.nm_last_error.argtypes = [vp]
    L.nm_create_note.restype = C.c_char_p
    L.nm_create_note.argtypes = [vp]
    L.nm_nslots.argtypes = [vp]
    L.nm_natoms.argtypes = [vp]
    L.nm_cus_per_replica.argtypes = [vp]
    L.nm_heal_count.argtypes = [vp]
    L.nm_get_const.argtypes = [vp, c_double_p, c_double_p]
    L.nm_set_state.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    L.nm_get_state.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    L.nm_set_thermo.argtypes = [vp, C.c_int, C.c_int, c_double_p]
    L.nm_get_slots.argtypes = [vp, C.c_int, c_int_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.nm_set_slots.argtypes = [vp, C.c_int, c_int_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.nm_init_lattice.argtypes = [vp, C.c_double, C.c_double, C.c_int]
    L.nm_lattice_state.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_uint32, C.c_int, C.c_double, C.c_int, c_double_p, c_double_p]
    L.nm_set_step.argtypes = [vp, C.c_uint32]
    L.nm_run_block.argtypes = [vp, C.c_int]
    L.nm_run_cycles.argtypes = [vp, C.c_int, C.c_int]
    L.nm_run_md.argtypes = [vp, C.c_int]
    L.nm_get_thermo.argtypes = [vp, c_double_p]
    L.nm_adapt.argtypes = [vp]
    L.nm_snapshot.argtypes = [vp]
    L.nm_snapshot_fetch.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    L.nm_exchange.argtypes = [vp, c_int_p]
    L.nm_synchronize.argtypes = [vp]
    L.nm_get_status.argtypes = [vp, c_int_p]
    L.nm_format_thrm.argtypes = [c_double_p, C.c_char_p, C.c_int]
    L.nm_format_traj.argtypes = [C.c_int, C.c_double, c_double_p, C.c_char_p, C.c_int]
    L.nm_append_outputs.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), c_double_p, c_double_p, c_double_p,
                                    C.c_int]
    L.nm_timing_reset.argtypes = [vp]
    L.nm_timing_get.argtypes = [vp, c_int_p, c_double_p]
    L.nm_stats_get.argtypes = [vp, c_double_p, C.c_int]
    L.nm_eval.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    L.nm_set_rng_tape.argtypes = [vp, c_double_p, c_int_p]
    L.nm_set_exchange_tape.argtypes = [vp, c_double_p, C.c_int]
    L.nm_set_trace.argtypes = [vp, C.c_int]
    L.nm_get_trace.argtypes = [vp, c_double_p, C.c_int]
    L.nm_get_perm.argtypes = [vp, c_int_p]
    L.nm_set_counters.argtypes = [vp, c_double_p, c_float_p]
    L.nm_get_exchange_crit.argtypes = [vp, c_double_p, C.c_int]
    for s in SYMBOLS:
        if s not in ('nm_last_error', 'nm_create_note'):
            getattr(L, s).restype = C.c_int
    L.nm_distr_histograms.restype = C.c_int
    L.nm_distr_histograms.argtypes = [C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int, c_double_p, C.c_int, c_double_p,
                                      c_float_p, c_float_p]
    L.nm_distr_last_error.restype = C.c_char_p
    c_long_p = C.POINTER(C.c_long)
    L.nm_parse_thrm.restype = C.c_int
    L.nm_parse_thrm.argtypes = [C.c_char_p, c_float_p, C.c_long, c_long_p, C.c_int]
    L.nm_parse_traj.restype = C.c_int
    L.nm_parse_traj.argtypes = [C.c_char_p, C.POINTER(C.c_uint16), c_float_p, c_float_p, C.c_long, C.c_long, c_long_p, c_long_p,
                                C.c_int]
    L.nm_parse_last_error.restype = C.c_char_p
    _lib = L
    return L
