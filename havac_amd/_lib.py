"""ctypes binding of libhavac_dev.so (include/havac_dev.h).

There is no fallback: if the HIP library has not been built, or cannot be
loaded, importing the symbols raises.  Nothing in this package computes SSV on
the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhavac_dev.so")

HAVAC_OK = 0
E_LENGTH, E_LOGIC, E_RUNTIME, E_NOMEM, E_HIT_OVERFLOW, E_NO_DEVICE, E_ARGUMENT, E_TIMEOUT = -1, -2, -3, -4, -5, -6, -7, -8

STATE_NEW, STATE_QUEUED, STATE_RUNNING, STATE_COMPLETED, STATE_ERROR = 1, 2, 3, 4, 5
STATE_ABORT, STATE_SUBMITTED, STATE_TIMEOUT, STATE_NORESPONSE = 6, 7, 8, 9

# every symbol include/havac_dev.h declares: (name, restype, argtypes)
_vp, _u8p, _i8p = C.c_void_p, C.c_void_p, C.c_void_p
SIGNATURES = {
    "havac_dev_create": (C.c_int, [C.c_uint32, C.POINTER(C.c_void_p)]),
    "havac_dev_create_multi": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_void_p)]),
    "havac_dev_device_count": (C.c_uint32, [_vp]),
    "havac_dev_destroy": (None, [_vp]),
    "havac_dev_set_hit_capacity": (C.c_int, [_vp, C.c_uint64]),
    "havac_dev_set_tuning": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_uint32]),
    "havac_dev_write_sequence": (C.c_int, [_vp, _u8p, C.c_uint64]),
    "havac_dev_write_phmm": (C.c_int, [_vp, _i8p, C.c_uint64]),
    "havac_dev_write_separator_mask": (C.c_int, [_vp, _u8p, C.c_uint64]),
    "havac_dev_write_sequence_chars": (C.c_int, [_vp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "havac_dev_read_sequence": (C.c_int, [_vp, C.c_void_p, C.c_uint64]),
    "havac_dev_read_separator_mask": (C.c_int, [_vp, C.c_void_p, C.c_uint64]),
    "havac_dev_write_sequence_records": (C.c_int, [_vp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "havac_dev_append_reverse_strand": (C.c_int, [_vp, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    "havac_dev_run_async": (C.c_int, [_vp]),
    "havac_dev_set_pipeline_depth": (C.c_int, [_vp, C.c_uint32]),
    "havac_dev_pipeline_depth": (C.c_uint32, [_vp]),
    "havac_dev_retire": (C.c_int, [_vp]),
    "havac_dev_open_runs": (C.c_uint32, [_vp]),
    "havac_dev_state": (C.c_int, [_vp]),
    "havac_dev_wait": (C.c_int, [_vp, C.c_uint32]),
    "havac_dev_abort": (C.c_int, [_vp]),
    "havac_dev_num_hits": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "havac_dev_read_hits": (C.c_int, [_vp, C.c_void_p, C.c_uint32]),
    "havac_dev_num_hits64": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "havac_dev_read_hits64": (C.c_int, [_vp, C.c_void_p, C.c_uint64]),
    "havac_dev_last_run_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "havac_dev_last_error": (C.c_char_p, [_vp]),
    "havac_ssv_ctx_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "havac_ssv_ctx_destroy": (None, [_vp]),
    "havac_ssv_enqueue": (C.c_int, [_vp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "havac_ssv_shard_window": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "havac_ssv_set_sequence_window": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "havac_ssv_set_separator_mask": (C.c_int, [_vp, C.c_void_p]),
    "havac_ssv_set_order_stream": (C.c_int, [_vp, C.c_void_p]),
    "havac_ssv_set_cell_trace": (C.c_int, [_vp, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]),
    "havac_ssv_finish": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "havac_ssv_finish_begin": (C.c_int, [_vp]),
    "havac_ssv_finish_end": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "havac_ssv_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "havac_ssv_set_split_tuning": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "havac_ssv_plan": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, _vp]),
    "havac_ssv_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "havac_ssv_last_kernel_variant": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "havac_ssv_wave_slots": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "havac_ssv_last_ordering": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "havac_ssv_sort_hits": (C.c_int, [_vp, C.c_void_p, C.c_uint64, C.c_void_p]),
    "havac_ssv_check_order": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "havac_ssv_last_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "havac_ssv_shard_cells": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "havac_ssv_shard_columns": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "havac_ssv_query": (C.c_int, [_vp]),
    "havac_ssv_wait_inputs": (C.c_int, [_vp, C.c_int]),
    "havac_pipe_create": (C.c_int, [C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "havac_pipe_destroy": (None, [_vp]),
    "havac_pipe_submit": (C.c_int, [_vp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "havac_pipe_collect": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]),
    "havac_pipe_run": (C.c_int, [_vp, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "havac_pipe_poll": (C.c_int, [_vp]),
    "havac_pipe_wait_inputs": (C.c_int, [_vp, C.c_int]),
    "havac_pipe_depth": (C.c_uint32, [_vp]),
    "havac_pipe_in_flight": (C.c_uint32, [_vp]),
    "havac_pipe_used_two_streams": (C.c_int, [_vp]),
    "havac_pipe_streams_used": (C.c_int, [_vp]),
    "havac_pipe_context": (C.c_void_p, [_vp, C.c_int]),
    "havac_pipe_set_gather": (C.c_int, [_vp, _vp]),
    "havac_pipe_wait_gathers": (C.c_int, [_vp]),
    "havac_pipe_gather_times": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "havac_pipe_last_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "havac_pipe_release": (C.c_int, [_vp]),
    "havac_pipe_last_error": (C.c_char_p, [_vp]),
    "havac_gather_info": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "havac_gather_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "havac_gather_unique_id": (C.c_int, [C.c_void_p]),
    "havac_gather_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "havac_gather_counts": (C.c_int, [_vp, C.c_int64, C.c_void_p, C.c_void_p]),
    "havac_gather_records": (C.c_int, [_vp, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "havac_gather_set_deadline": (C.c_int, [_vp, C.c_uint32]),
    "havac_gather_wait": (C.c_int, [_vp]),
    "havac_gather_use_library": (C.c_int, [C.c_char_p]),
    "havac_gather_last_error": (C.c_char_p, [_vp]),
    "havac_gather_destroy": (None, [_vp]),
    "havac_ssv_ctx_last_error": (C.c_char_p, [_vp]),
    "havac_dev_version": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Load libhavac_dev.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with havac_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
