"""ctypes binding of libccsim.so (include/ccsim.h).  Plumbing only: marshals the numpy holders of
model.py into the C structs and calls the HIP engine.  There is no CPU fallback: if the library or
the GPU is missing, calls raise."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build
from . import model as M

MAX_RES = M.MAX_RES
MAX_LABEL_COLS = 32
NREASON = M.NREASON
XCHG_WORDS = 32
MODE_SEQUENTIAL, MODE_BATCHED = 0, 1
MODES = {"sequential": MODE_SEQUENTIAL, "batched": MODE_BATCHED}

_p64 = C.POINTER(C.c_int64)
_p32 = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class CConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("rounds_per_sync", C.c_int32), ("use_graph", C.c_int32), ("time_passes", C.c_int32)]


class CNodes(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64), ("global_offset", C.c_int64), ("n_global", C.c_int64), ("n_scalar", C.c_int32),
        ("alloc", _p64 * MAX_RES), ("alloc_pods", _p32), ("req", _p64 * MAX_RES), ("nz_mcpu", _p64), ("nz_mem", _p64),
        ("pod_count", _p32), ("taintset_id", _p32), ("unschedulable", _pu8), ("n_label_cols", C.c_int32),
        ("label_cols", _p32 * MAX_LABEL_COLS),
    ]


class CReq(C.Structure):
    _fields_ = [("col", C.c_int32), ("table_off", C.c_int32)]


class CTerm(C.Structure):
    _fields_ = [("first_req", C.c_int32), ("n_req", C.c_int32), ("weight", C.c_int32)]


class CSpread(C.Structure):
    _fields_ = [("col", C.c_int32), ("max_skew", C.c_int32), ("min_domains", C.c_int32), ("hard", C.c_int32),
                ("self_match", C.c_int32), ("n_domains", C.c_int32), ("node_match_count", _p32), ("node_included", _pu8),
                ("is_hostname", C.c_int32), ("missing_value", C.c_int32)]


class CIpa(C.Structure):
    _fields_ = [
        ("n_keys", C.c_int32), ("key_col", C.c_int32 * M.MAX_IPA_KEYS), ("key_ndom", C.c_int32 * M.MAX_IPA_KEYS),
        ("n_aff_terms", C.c_int32), ("aff_key", C.c_int32 * M.MAX_IPA_TERMS), ("self_aff", C.c_int32), ("aff_existing", _p32),
        ("n_anti_terms", C.c_int32), ("anti_key", C.c_int32 * M.MAX_IPA_TERMS), ("anti_self", C.c_int32 * M.MAX_IPA_TERMS),
        ("anti_existing", _p32 * M.MAX_IPA_TERMS), ("exist_anti", _p32 * M.MAX_IPA_KEYS),
        ("score_existing", _p64 * M.MAX_IPA_KEYS), ("score_self", C.c_int64 * M.MAX_IPA_KEYS), ("entries_existing", C.c_int64),
        ("self_entries", C.c_int32 * M.MAX_IPA_KEYS),
    ]


class CPod(C.Structure):
    _fields_ = [
        ("req", C.c_int64 * MAX_RES), ("has_scalar_entries", C.c_int32), ("nz_mcpu", C.c_int64), ("nz_mem", C.c_int64),
        ("n_taintsets", C.c_int32), ("taint_filter_ok", _pu8), ("taint_prefer_cnt", _p32),
        ("tolerates_unschedulable", C.c_int32), ("affinity_filter_active", C.c_int32), ("has_node_selector", C.c_int32),
        ("node_selector", CTerm), ("has_required_terms", C.c_int32), ("n_required", C.c_int32),
        ("required", C.POINTER(CTerm)), ("n_preferred", C.c_int32), ("preferred", C.POINTER(CTerm)),
        ("n_reqs", C.c_int32), ("reqs", C.POINTER(CReq)), ("req_tables_len", C.c_int64), ("req_tables", _pu8),
        ("n_spread", C.c_int32), ("spread", CSpread * M.MAX_TSC), ("has_ipa", C.c_int32), ("ipa", CIpa),
        ("has_host_ports", C.c_int32), ("host_ports_conflict", _pu8), ("image_score", _pu8),
        ("volume_exclusive", C.c_int32), ("volume_veto", _pu8),
    ]


class CProfile(C.Structure):
    _fields_ = [
        ("filter_mask", C.c_uint32), ("w_taint", C.c_int32), ("w_nodeaffinity", C.c_int32), ("w_fit", C.c_int32),
        ("w_balanced", C.c_int32), ("w_topologyspread", C.c_int32), ("w_interpodaffinity", C.c_int32), ("n_fit_res", C.c_int32),
        ("fit_res", C.c_int32 * MAX_RES), ("fit_res_w", C.c_int64 * MAX_RES), ("n_bal_res", C.c_int32),
        ("bal_res", C.c_int32 * MAX_RES), ("percentage_of_nodes_to_score", C.c_int32), ("w_imagelocality", C.c_int32),
    ]


class CReport(C.Structure):
    _fields_ = [
        ("placed", C.c_int64), ("stop", C.c_int32), ("per_node_count", _p32), ("per_node_cap", C.c_int64),
        ("log", _p32), ("log_cap", C.c_int64), ("log_len", C.c_int64), ("hist", C.c_int64 * NREASON),
        ("hist_taintset", _p64), ("hist_taintset_cap", C.c_int32), ("n_code_unschedulable", C.c_int64),
        ("rounds", C.c_int64), ("scans", C.c_int64), ("evaluated_total", C.c_int64), ("last_feasible", C.c_int32),
        ("kernel_ns", C.c_int64), ("pass_kernel_ns", C.c_int64), ("pass_launches", C.c_int64), ("bytes_per_scan", C.c_int64),
        ("per_spec_count", _p32), ("per_spec_cap", C.c_int32), ("stop_spec", C.c_int32),
        ("per_node_count_narrow", C.c_void_p), ("per_node_narrow_width", C.c_int32), ("per_node_filled_width", C.c_int32),
    ]


class CCycle(C.Structure):
    _fields_ = [("node", C.c_int64), ("evaluated_nodes", C.c_int32), ("feasible_nodes", C.c_int32)]


ABI_VERSION = 5  # CCSIM_ABI_VERSION of include/ccsim.h

# every symbol include/ccsim.h declares (checked by tests/test_abi.py without a GPU)
SYMBOLS = {
    "ccsim_abi_version": (C.c_int32, []),
    "ccsim_create": (C.c_int, [C.POINTER(CConfig), C.POINTER(C.c_void_p)]),
    "ccsim_destroy": (None, [C.c_void_p]),
    "ccsim_last_error": (C.c_char_p, [C.c_void_p]),
    "ccsim_load_nodes": (C.c_int, [C.c_void_p, C.POINTER(CNodes)]),
    "ccsim_set_profile": (C.c_int, [C.c_void_p, C.POINTER(CProfile)]),
    "ccsim_set_pod": (C.c_int, [C.c_void_p, C.POINTER(CPod)]),
    "ccsim_set_pods": (C.c_int, [C.c_void_p, C.POINTER(CPod), C.c_int32]),
    "ccsim_schedule_pod": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(CCycle)]),
    "ccsim_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(CReport)]),
    "ccsim_schedule_one": (C.c_int, [C.c_void_p, C.POINTER(CCycle)]),
    "ccsim_read_state": (C.c_int, [C.c_void_p, _p64, _p64, _p64, _p64, _p32]),
    "ccsim_dist_begin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "ccsim_dist_scan": (C.c_int, [C.c_void_p]),
    "ccsim_dist_decide": (C.c_int, [C.c_void_p]),
    "ccsim_dist_poll": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ccsim_dist_finish": (C.c_int, [C.c_void_p, C.POINTER(CReport)]),
    "ccsim_dist_table_count": (C.c_int, [C.c_void_p]),
    "ccsim_dist_table": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32)]),
    "ccsim_dist_tables_done": (C.c_int, [C.c_void_p]),
    "ccsim_dist_unique_id": (C.c_int, [_pu8]),
    "ccsim_dist_comm_init": (C.c_int, [C.c_void_p, _pu8, C.c_int32, C.c_int32]),
    "ccsim_dist_sync_tables": (C.c_int, [C.c_void_p]),
    "ccsim_dist_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ccsim_dist_cw_eligible": (C.c_int, [C.c_void_p]),
    "ccsim_dist_cw_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccsim_dist_cw_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "ccsim_dist_cw_scan": (C.c_int, [C.c_void_p]),
    "ccsim_dist_cw_decide": (C.c_int, [C.c_void_p]),
    "ccsim_dist_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(CReport)]),
    "ccsim_dist_mbox_info": (C.c_int, [C.c_void_p, _pu8]),
    "ccsim_dist_mbox_connect": (C.c_int, [C.c_void_p, _pu8, C.c_int32, C.c_int32]),
    "ccsim_dist_mbox_eligible": (C.c_int, [C.c_void_p]),
    "ccsim_dist_mbox_launch": (C.c_int, [C.c_void_p]),
    "ccsim_dist_mbox_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "ccsim_dist_mbox_finish": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccsim_reset_state": (C.c_int, [C.c_void_p]),
    "ccsim_host_alloc": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "ccsim_host_free": (None, [C.c_void_p, C.c_void_p]),
    "ccsim_time_scan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ccsim_debug_persist_prof": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ccsim_debug_multi_stops": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ccsim_debug_multi_memo": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ccsim_debug_coupled": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ccsim_debug_sampled": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ccsim_debug_dist": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
}

_lib = None


def load(build_if_missing: bool = True):
    """dlopen libccsim.so and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is None:
        path = os.environ.get("CCSIM_LIB") or _build.lib_path()  # CCSIM_LIB: another build of the same ABI (A/B timing runs)
        if not os.path.exists(path):
            if not build_if_missing:
                raise FileNotFoundError(path)
            _build.build_all()
        # One HIP runtime per process: torch ships its own libamdhip64.so.7.  Loading torch FIRST makes the
        # loader resolve libccsim's libamdhip64.so.7 dependency to that same copy, so torch streams and
        # tensors (multi-GPU path) and the engine share one runtime.  (Engine first, torch second loads two
        # runtimes and torch then sees no GPU.)  C / cgo consumers simply get /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if lib.ccsim_abi_version() != ABI_VERSION:
            raise RuntimeError("libccsim ABI version mismatch")
        _lib = lib
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


class CcsimError(RuntimeError):
    rc = 0  # the entry point's return code (e.g. -38 = -ENOSYS: a shape this form of the engine does not take)


DIST_ID_BYTES = 128
MBOX_INFO_BYTES = 96


def dist_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0; hand the bytes to every rank, e.g. torch.distributed.broadcast_object_list)."""
    buf = (C.c_uint8 * DIST_ID_BYTES)()
    rc = load().ccsim_dist_unique_id(buf)
    if rc != 0:
        raise CcsimError(f"ccsim_dist_unique_id failed rc={rc} (librccl.so.1 not loadable?)")
    return bytes(buf)


def marshal_nodes(nodes: M.NodesSoA, keep: list, global_offset: int = 0, n_global: Optional[int] = None) -> CNodes:
    s = CNodes()
    s.n_nodes = nodes.n
    s.global_offset = int(global_offset)
    s.n_global = int(n_global if n_global is not None else nodes.n)
    s.n_scalar = nodes.n_scalar

    def arr(a, dt, t):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return _ptr(a, t)

    for c in range(len(nodes.alloc)):
        s.alloc[c] = arr(nodes.alloc[c], np.int64, _p64)
        s.req[c] = arr(nodes.req[c], np.int64, _p64)
    s.alloc_pods = arr(nodes.alloc_pods, np.int32, _p32)
    s.nz_mcpu = arr(nodes.nz_mcpu, np.int64, _p64)
    s.nz_mem = arr(nodes.nz_mem, np.int64, _p64)
    s.pod_count = arr(nodes.pod_count, np.int32, _p32)
    s.taintset_id = arr(nodes.taintset_id, np.int32, _p32)
    s.unschedulable = arr(nodes.unschedulable, np.uint8, _pu8)
    assert len(nodes.label_cols) <= MAX_LABEL_COLS
    s.n_label_cols = len(nodes.label_cols)
    for i, col in enumerate(nodes.label_cols):
        s.label_cols[i] = arr(col, np.int32, _p32)
    return s


def marshal_pod(pod: M.PodSpec, keep: list) -> CPod:
    s = CPod()
    for c, v in enumerate(np.asarray(pod.req, dtype=np.int64)):
        s.req[c] = int(v)
    s.has_scalar_entries = int(bool(pod.has_scalar_entries))
    s.nz_mcpu, s.nz_mem = int(pod.nz_mcpu), int(pod.nz_mem)
    ok = np.ascontiguousarray(pod.taint_filter_ok, dtype=np.uint8)
    cnt = np.ascontiguousarray(pod.taint_prefer_cnt, dtype=np.int32)
    keep += [ok, cnt]
    s.n_taintsets = int(ok.shape[0])
    s.taint_filter_ok = _ptr(ok, _pu8)
    s.taint_prefer_cnt = _ptr(cnt, _p32)
    s.tolerates_unschedulable = int(bool(pod.tolerates_unschedulable))
    s.affinity_filter_active = int(bool(pod.affinity_filter_active))
    reqs, tables, off = [], [], 0

    def add_term(term_reqs, weight=0):
        nonlocal off
        first = len(reqs)
        for col, table in term_reqs:
            t = np.ascontiguousarray(table, dtype=np.uint8)
            reqs.append(CReq(int(col), off))
            tables.append(t)
            off += int(t.shape[0])
        return CTerm(first, len(term_reqs), int(weight))

    s.has_node_selector = int(bool(pod.has_node_selector))
    s.node_selector = add_term(pod.node_selector)
    s.has_required_terms = int(bool(pod.has_required_terms))
    rt = [add_term(t) for t in pod.required]
    pt = [add_term(t, w) for (w, t) in pod.preferred]
    s.n_required, s.n_preferred, s.n_reqs = len(rt), len(pt), len(reqs)
    rta = (CTerm * max(1, len(rt)))(*rt)
    pta = (CTerm * max(1, len(pt)))(*pt)
    rqa = (CReq * max(1, len(reqs)))(*reqs)
    tab = np.concatenate(tables) if tables else np.zeros(1, np.uint8)
    keep += [rta, pta, rqa, tab]
    s.required = C.cast(rta, C.POINTER(CTerm))
    s.preferred = C.cast(pta, C.POINTER(CTerm))
    s.reqs = C.cast(rqa, C.POINTER(CReq))
    s.req_tables_len = int(tab.shape[0])
    s.req_tables = _ptr(tab, _pu8)
    spread = list(getattr(pod, "spread", None) or [])
    if len(spread) > M.MAX_TSC:
        raise CcsimError("too many topology spread constraints")
    s.n_spread = len(spread)
    for i, k in enumerate(spread):
        c = s.spread[i]
        c.col, c.max_skew, c.min_domains = int(k.col), int(k.max_skew), int(k.min_domains)
        c.hard, c.self_match, c.n_domains = int(bool(k.hard)), int(bool(k.self_match)), int(k.n_domains)
        c.is_hostname = int(bool(k.is_hostname))
        c.missing_value = int(getattr(k, "missing_value", 0))
        if k.node_match_count is not None:
            a = np.ascontiguousarray(k.node_match_count, dtype=np.int32)
            keep.append(a)
            c.node_match_count = _ptr(a, _p32)
        if k.node_included is not None:
            a = np.ascontiguousarray(k.node_included, dtype=np.uint8)
            keep.append(a)
            c.node_included = _ptr(a, _pu8)
    ipa = getattr(pod, "ipa", None)
    s.has_ipa = int(ipa is not None)
    if ipa is not None:
        _fill_ipa(s.ipa, ipa, keep)
    s.has_host_ports = int(bool(getattr(pod, "has_host_ports", False)))
    s.volume_exclusive = int(bool(getattr(pod, "volume_exclusive", False)))
    for name in ("host_ports_conflict", "image_score", "volume_veto"):
        a = getattr(pod, name, None)
        if a is not None:
            a = np.ascontiguousarray(a, dtype=np.uint8)
            keep.append(a)
            setattr(s, name, _ptr(a, _pu8))
    return s


def _fill_ipa(c, ipa: M.InterPodAffinity, keep: list):
    def arr(a, dt, t):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return _ptr(a, t)

    if len(ipa.key_cols) > M.MAX_IPA_KEYS or max(len(ipa.aff_keys), len(ipa.anti_keys)) > M.MAX_IPA_TERMS:
        raise CcsimError("too many inter-pod affinity keys / terms")
    c.n_keys = len(ipa.key_cols)
    for k, (col, nd) in enumerate(zip(ipa.key_cols, ipa.key_ndom)):
        c.key_col[k], c.key_ndom[k] = int(col), int(nd)
    c.n_aff_terms = len(ipa.aff_keys)
    for t, k in enumerate(ipa.aff_keys):
        c.aff_key[t] = int(k)
    c.self_aff = int(bool(ipa.self_aff))
    if ipa.aff_existing is not None:
        c.aff_existing = arr(ipa.aff_existing, np.int32, _p32)
    c.n_anti_terms = len(ipa.anti_keys)
    for t, k in enumerate(ipa.anti_keys):
        c.anti_key[t] = int(k)
        c.anti_self[t] = int(bool(ipa.anti_self[t]))
        if ipa.anti_existing and ipa.anti_existing[t] is not None:
            c.anti_existing[t] = arr(ipa.anti_existing[t], np.int32, _p32)
    for k in range(c.n_keys):
        if ipa.exist_anti and ipa.exist_anti[k] is not None:
            c.exist_anti[k] = arr(ipa.exist_anti[k], np.int32, _p32)
        if ipa.score_existing and ipa.score_existing[k] is not None:
            c.score_existing[k] = arr(ipa.score_existing[k], np.int64, _p64)
        c.score_self[k] = int(ipa.score_self[k]) if ipa.score_self else 0
        c.self_entries[k] = int(ipa.self_entries[k]) if ipa.self_entries else 0
    c.entries_existing = int(ipa.entries_existing)


def marshal_profile(p: M.Profile) -> CProfile:
    s = CProfile()
    s.filter_mask = int(p.filter_mask)
    s.w_taint, s.w_nodeaffinity, s.w_fit = int(p.w_taint), int(p.w_nodeaffinity), int(p.w_fit)
    s.w_balanced, s.w_topologyspread = int(p.w_balanced), int(p.w_topologyspread)
    s.w_interpodaffinity = int(p.w_interpodaffinity)
    s.n_fit_res = len(p.fit_res)
    for i, (c, w) in enumerate(zip(p.fit_res, p.fit_res_w)):
        s.fit_res[i], s.fit_res_w[i] = int(c), int(w)
    s.n_bal_res = len(p.bal_res)
    for i, c in enumerate(p.bal_res):
        s.bal_res[i] = int(c)
    s.percentage_of_nodes_to_score = int(p.percentage_of_nodes_to_score)
    s.w_imagelocality = int(p.w_imagelocality)
    return s


class Engine:
    """One engine = one GPU's shard of the snapshot (ccsim_engine*)."""

    def __init__(self, device: int = 0, stream: int = 0, rounds_per_sync: int = 0, use_graph: bool = True,
                 time_passes: bool = False):
        self.lib = load()
        # stream: a hipStream_t handle (e.g. torch.cuda.Stream().cuda_stream); 0/None = the engine creates its own
        cfg = CConfig(ABI_VERSION, int(device), C.c_void_p(stream) if stream else None, int(rounds_per_sync), int(use_graph),
                      int(time_passes))
        h = C.c_void_p()
        rc = self.lib.ccsim_create(C.byref(cfg), C.byref(h))
        if rc != 0 or not h:
            raise CcsimError(f"ccsim_create failed rc={rc} (is a HIP device visible?)")
        self.h = h
        self.n = 0
        self.n_taintsets = 1
        self.n_pods = 1

    def close(self):
        if getattr(self, "h", None):
            self._free_pinned()
            self.lib.ccsim_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.ccsim_last_error(self.h)
            err = CcsimError(f"{what} failed rc={rc}: {msg.decode() if msg else ''}")
            err.rc = rc
            raise err

    def load(self, nodes: M.NodesSoA, pod, profile: M.Profile, global_offset: int = 0,
             n_global: Optional[int] = None):
        """`pod`: one M.PodSpec, or a list of them (cycled round-robin: ccsim_set_pods)."""
        if not isinstance(pod, (list, tuple)) and getattr(pod, "soft_relaxed", False):
            if global_offset or (n_global not in (None, nodes.n)):
                raise ValueError("a pod scored with requireAllTopologies = false: apply model.relax_soft to the WHOLE snapshot before sharding it")
            nodes, pod = M.relax_soft(nodes, pod)  # (the engine form: one more value id for the nodes that lack a key, ccsim.h missing_value)
        keep: list = []
        self._chk(self.lib.ccsim_load_nodes(self.h, C.byref(marshal_nodes(nodes, keep, global_offset, n_global))), "ccsim_load_nodes")
        self.n = nodes.n
        self.set_profile(profile)
        if isinstance(pod, (list, tuple)):
            self.set_pods(pod)
        else:
            self.set_pod(pod)

    def set_profile(self, profile: M.Profile):
        self._chk(self.lib.ccsim_set_profile(self.h, C.byref(marshal_profile(profile))), "ccsim_set_profile")

    def set_pod(self, pod: M.PodSpec):
        if getattr(pod, "soft_relaxed", False):
            raise ValueError("a pod scored with requireAllTopologies = false needs its nodes' label columns extended: Engine.load / model.relax_soft")
        keep: list = []
        cp = marshal_pod(pod, keep)
        self._chk(self.lib.ccsim_set_pod(self.h, C.byref(cp)), "ccsim_set_pod")
        self.n_taintsets = int(cp.n_taintsets)

    def set_pods(self, pods):
        """Several pod specs, cycled round-robin by run() (include/ccsim.h ccsim_set_pods)."""
        keep: list = []
        arr = (CPod * len(pods))()
        for i, p in enumerate(pods):
            arr[i] = marshal_pod(p, keep)
        self._chk(self.lib.ccsim_set_pods(self.h, arr, len(pods)), "ccsim_set_pods")
        self.n_taintsets = max(int(a.n_taintsets) for a in arr)
        self.n_pods = len(pods)

    def schedule_pod(self, pod_idx: int):
        cyc = CCycle()
        self._chk(self.lib.ccsim_schedule_pod(self.h, int(pod_idx), C.byref(cyc)), "ccsim_schedule_pod")
        return int(cyc.node), int(cyc.evaluated_nodes), int(cyc.feasible_nodes)

    def _pinned_per_node(self):
        """One page-locked int32[n] per engine (ccsim_host_alloc), reused by every run(reuse_buffers=True)."""
        n = max(1, self.n)
        buf = getattr(self, "_pin_per_node", None)
        if buf is None or buf[1].shape[0] != n:
            self._free_pinned()
            ptr = self.lib.ccsim_host_alloc(self.h, n * 4)
            if not ptr:
                raise MemoryError("ccsim_host_alloc failed")
            arr = np.ctypeslib.as_array((C.c_int32 * n).from_address(ptr))
            self._pin_per_node = buf = (ptr, arr)
        return buf[1]

    def _free_pinned(self):
        buf = getattr(self, "_pin_per_node", None)
        # the last result handed out with reuse_buffers=True holds a VIEW of the page-locked array: give it a copy of its own
        # before the memory goes away (close(), a snapshot of another size), so that keeping a result is never a use-after-free
        for ref in getattr(self, "_pin_results", []):
            res = ref()
            if res is not None and buf is not None and res.per_node_count is not None and res.per_node_count.base is not None:
                res.per_node_count = np.array(res.per_node_count, copy=True)
        self._pin_results = []
        self._rep_cache = None
        if buf is not None and self.h:
            self.lib.ccsim_host_free(self.h, buf[0])
        self._pin_per_node = None
        nb = getattr(self, "_pin_narrow", None)
        if nb is not None and self.h:
            self.lib.ccsim_host_free(self.h, nb[0])
        self._pin_narrow = None

    def _pinned_narrow(self):
        """Page-locked bytes for ccsim_report.per_node_count_narrow (ABI 5): n elements of up to two bytes."""
        n = max(1, self.n)
        nb = getattr(self, "_pin_narrow", None)
        if nb is None or nb[1].shape[0] != 2 * n:
            if nb is not None and self.h:
                self.lib.ccsim_host_free(self.h, nb[0])
            ptr = self.lib.ccsim_host_alloc(self.h, 2 * n)
            if not ptr:
                raise MemoryError("ccsim_host_alloc failed")
            self._pin_narrow = nb = (ptr, np.ctypeslib.as_array((C.c_uint8 * (2 * n)).from_address(ptr)))
        return nb

    def _report(self, want_log: bool, log_cap: int, reuse_buffers: bool = False):
        if reuse_buffers and not want_log:
            # a caller that simulates repeatedly: the report block and its small arrays are built once (the binding's own work was
            # ~14 of a 290 us step at 1M nodes); the library zeroes the histograms it fills
            key = (self.n, self.n_taintsets, self.n_pods, id(getattr(self, "_pin_per_node", None)))
            c = getattr(self, "_rep_cache", None)
            if c is not None and c[0] == key and c[2] is self._pinned_per_node():
                c[1].stop_spec = -1
                self._per_spec = c[4]  # (a run with a log in between built its own block and arrays: the cached block points at THESE)
                return c[1], c[2], None, c[3]
        rep = CReport()
        per_node = self._pinned_per_node() if reuse_buffers else np.zeros(max(1, self.n), np.int32)
        rep.per_node_count = _ptr(per_node, _p32)
        rep.per_node_cap = per_node.shape[0]
        log = None
        if want_log:
            log = np.full(max(1, int(log_cap)), -1, np.int32)
            rep.log = _ptr(log, _p32)
            rep.log_cap = log.shape[0]
        ht = np.zeros(max(1, self.n_taintsets), np.int64)
        rep.hist_taintset = _ptr(ht, _p64)
        rep.hist_taintset_cap = ht.shape[0]
        self._per_spec = np.zeros(max(1, self.n_pods), np.int32)
        rep.per_spec_count = _ptr(self._per_spec, _p32)
        rep.per_spec_cap = self._per_spec.shape[0]
        rep.stop_spec = -1
        if reuse_buffers and not want_log:
            # (the tuple keeps every array the block points at alive for as long as the block can be handed out again)
            self._rep_cache = ((self.n, self.n_taintsets, self.n_pods, id(self._pin_per_node)), rep, per_node, ht, self._per_spec)
        return rep, per_node, log, ht

    def _result(self, rep, per_node, log, ht, reuse_buffers: bool = False) -> M.RunResult:
        return M.RunResult(
            placed=int(rep.placed), stop=int(rep.stop), per_node_count=per_node[: self.n] if reuse_buffers else per_node[: self.n].copy(),
            log=log[: int(rep.log_len)].copy() if log is not None else None,
            hist=np.frombuffer(rep.hist, dtype=np.int64).copy(), hist_taintset=ht.copy(),
            n_code_unschedulable=int(rep.n_code_unschedulable), rounds=int(rep.rounds),
            evaluated_total=int(rep.evaluated_total), last_feasible=int(rep.last_feasible), scans=int(rep.scans),
            kernel_ns=int(rep.kernel_ns), pass_kernel_ns=int(rep.pass_kernel_ns), pass_launches=int(rep.pass_launches), bytes_per_scan=int(rep.bytes_per_scan),
            per_spec_count=self._per_spec[: self.n_pods].copy(), stop_spec=int(rep.stop_spec),
        )

    def run(self, max_limit: int = 0, mode: str = "sequential", want_log: bool = True, log_cap: Optional[int] = None,
            reuse_buffers: bool = False, narrow_counts: int = 0) -> M.RunResult:
        """`narrow_counts` (1 or 2, with `reuse_buffers`): offer ccsim_report.per_node_count_narrow (ABI 5) -- when the run's form
        supports it and every count fits, `per_node_count` of the result is a uint8 / uint16 view instead of int32 (same values; a
        quarter / half of the bytes over PCIe).
        `reuse_buffers`: the per-node counts land in the engine's page-locked result array (ccsim_host_alloc) and the result
        holds a VIEW of it: its CONTENTS are valid until the next run of this engine overwrites them -- what a caller that
        simulates repeatedly does with its own arrays.  When the array itself is released (close(), a snapshot of another size)
        every such result still alive is given a private copy first: keeping a result past close() is never a use-after-free."""
        if log_cap is None:
            log_cap = max_limit if max_limit > 0 else 1 << 22
        rep, per_node, log, ht = self._report(want_log, log_cap, reuse_buffers)
        rep.per_node_count_narrow, rep.per_node_narrow_width, rep.per_node_filled_width = None, 0, 0
        if narrow_counts in (1, 2) and reuse_buffers:
            nb = self._pinned_narrow()
            rep.per_node_count_narrow, rep.per_node_narrow_width = nb[0], int(narrow_counts)
        self._chk(self.lib.ccsim_run(self.h, int(max_limit), MODES[mode], C.byref(rep)), "ccsim_run")
        if rep.per_node_filled_width in (1, 2):
            per_node = self._pin_narrow[1].view(np.uint8 if rep.per_node_filled_width == 1 else np.uint16)
        res = self._result(rep, per_node, log, ht, reuse_buffers)
        if reuse_buffers:
            import weakref

            self._pin_results = [r for r in getattr(self, "_pin_results", []) if r() is not None] + [weakref.ref(res)]
        return res

    def schedule_one(self):
        cyc = CCycle()
        self._chk(self.lib.ccsim_schedule_one(self.h, C.byref(cyc)), "ccsim_schedule_one")
        return int(cyc.node), int(cyc.evaluated_nodes), int(cyc.feasible_nodes)

    def read_state(self):
        n = max(1, self.n)
        out = [np.zeros(n, np.int64) for _ in range(4)] + [np.zeros(n, np.int32)]
        self._chk(self.lib.ccsim_read_state(self.h, *[_ptr(a, _p64) for a in out[:4]], _ptr(out[4], _p32)), "ccsim_read_state")
        return dict(req_mcpu=out[0][: self.n], req_mem=out[1][: self.n], nz_mcpu=out[2][: self.n], nz_mem=out[3][: self.n],
                    pod_count=out[4][: self.n])

    def reset_state(self):
        self._chk(self.lib.ccsim_reset_state(self.h), "ccsim_reset_state")

    def time_scan(self, iters: int, mode: str = "sequential"):
        ns, by = C.c_int64(), C.c_int64()
        self._chk(self.lib.ccsim_time_scan(self.h, MODES[mode], int(iters), C.byref(ns), C.byref(by)), "ccsim_time_scan")
        return int(ns.value), int(by.value)

    def multi_stops(self):
        out = (C.c_int64 * 8)()
        self._chk(self.lib.ccsim_debug_multi_stops(self.h, out), "ccsim_debug_multi_stops")
        return [int(x) for x in out]

    def multi_memo(self):
        """The multi-spec score memo of the last run (ccsim_debug_multi_memo): on?, pod-scans read from it / computed, bytes."""
        out = (C.c_int64 * 8)()
        self._chk(self.lib.ccsim_debug_multi_memo(self.h, out), "ccsim_debug_multi_memo")
        d = {"on": bool(out[0]), "memo_scans": int(out[1]), "full_scans": int(out[2]), "bytes": int(out[3])}
        if out[7]:  # scan workgroup (0, 0), microseconds per scan: loads + staging, evaluation, merge
            d["scan_us"] = [round(out[4 + i] / 100.0 / out[7], 2) for i in range(3)]
        return d

    def coupled_info(self):
        """How the last run of one template with topology-coupled plugins was resolved (ccsim_debug_coupled)."""
        out = (C.c_int64 * 16)()
        self._chk(self.lib.ccsim_debug_coupled(self.h, out), "ccsim_debug_coupled")
        d = {"plan": bool(out[0]), "windows": int(out[1]), "fell_back": bool(out[2]), "window": int(out[3]), "list_len": int(out[4]),
             "fast_windows": int(out[5]), "full_windows": int(out[6]), "swept": int(out[7])}
        if out[15]:  # CCSIM_CW_PROF=1: microseconds per cycle of the deciding wave, by phase
            names = ["stage", "setup", "verdicts", "raw_scores", "argmax", "commit", "write_back"]
            d["prof_us_per_cycle"] = {k: round(out[8 + i] / 100.0 / out[15], 3) for i, k in enumerate(names)}
            d["cycles"] = int(out[15])
        return d

    def dist_info(self):
        """Which form this engine's library-driven sharded runs took (ccsim_debug_dist)."""
        out = (C.c_int64 * 8)()
        self._chk(self.lib.ccsim_debug_dist(self.h, out), "ccsim_debug_dist")
        forms = {0: None, 1: "mailbox", 2: "passes", 3: "windows"}
        return {"mailboxes_connected": bool(out[0]), "mailbox_go": int(out[1]), "mailbox_abandoned": int(out[2]), "last_form": forms.get(int(out[3])),
                "mailbox_launches": int(out[4]), "ranks": int(out[5])}

    def sampled_info(self):
        """Which resident form the last sequential run took: the sampled search (percentageOfNodesToScore < 100) a lap / a cycle at a time, its zone form, or
        the full search on the same summaries (ccsim_debug_sampled)."""
        out = (C.c_int64 * 16)()
        self._chk(self.lib.ccsim_debug_sampled(self.h, out), "ccsim_debug_sampled")
        d = {"resident": bool(out[0]), "laps_form": out[1] in (1, 4), "handed_over_to_full_search": out[1] == 4, "zone_form": out[1] == 2, "full_search_form": out[1] == 3, "launches": int(out[2]), "laps": int(out[3]), "slow_stretches": int(out[4]),
             "block": 1 << int(out[5]), "blocks": int(out[6]), "K": int(out[7])}
        if out[3] and any(out[8:16]):  # CCSIM_SB_PROF=1: microseconds per lap (zone form: per cycle), by phase
            if d["full_search_form"]:  # per node change: ticks of four phases; counts
                ch = max(1, int(out[12]))
                d["prof"] = {"node_changes": int(out[12]), "block_changes": int(out[13]), "evaluations": int(out[15]),
                             "us_per_node_change": {k: round(out[8 + i] / 100.0 / ch, 3) for i, k in enumerate(["trip_issue_and_shadow_reductions", "wait_block_reduction", "stores", "streak_evaluations"])}}
                return d
            if d["zone_form"]:
                names = ["zone_mask_start_block", "ring_prefix_stop_block", "kept_nodes_wait", "placement_entry_next_mask", "w1_stop_block", "w1_entries", "w2_idle", "w2_entries"]
            else:
                names = ["cut_blocks_tree_range_queries", "decide", "wait_commit", "leaves_next_cuts", "w1_tree", "w1_range_queries", "commit_wave", "w0_next_cuts"]
            d["prof_us_per_lap"] = {k: round(out[8 + i] / 100.0 / out[3], 3) for i, k in enumerate(names)}
        return d

    def persist_prof(self):
        """Phase breakdown of the last persistent batched launch (ccsim_debug_persist_prof), in microseconds."""
        out = (C.c_int64 * 16)()
        self._chk(self.lib.ccsim_debug_persist_prof(self.h, out), "ccsim_debug_persist_prof")
        names = ["scan_list", "plan", "apply", "block_reduce", "grid_reduce", "rescore"]
        d = {k: out[i] / 100.0 for i, k in enumerate(names)}
        d["levels"] = int(out[6])
        d["load"], d["rescore_maxima"], d["write_back"] = out[7] / 100.0, out[8] / 100.0, out[10] / 100.0
        return d

    # ---- distributed stepping (collective supplied by the caller, see dist.py) ----
    def dist_begin(self, max_limit: int, mode: str, n_ranks: int, rank: int, send_ptr: int, recv_ptr: int, log_cap: int = 0):
        self._chk(self.lib.ccsim_dist_begin(self.h, int(max_limit), MODES[mode], int(n_ranks), int(rank),
                                            C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(log_cap)), "ccsim_dist_begin")

    def dist_tables(self):
        """[(device pointer, element count, element bytes, op)] of the replicated topology tables (op 0 SUM, 1 MAX)."""
        out = []
        for i in range(self.lib.ccsim_dist_table_count(self.h)):
            ptr, n, eb, op = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_int32()
            self._chk(self.lib.ccsim_dist_table(self.h, i, C.byref(ptr), C.byref(n), C.byref(eb), C.byref(op)), "ccsim_dist_table")
            out.append((int(ptr.value), int(n.value), int(eb.value), int(op.value)))
        return out

    def dist_tables_done(self):
        self._chk(self.lib.ccsim_dist_tables_done(self.h), "ccsim_dist_tables_done")

    def dist_scan(self):
        self._chk(self.lib.ccsim_dist_scan(self.h), "ccsim_dist_scan")

    def dist_decide(self):
        self._chk(self.lib.ccsim_dist_decide(self.h), "ccsim_dist_decide")

    def dist_poll(self):
        done, placed = C.c_int32(), C.c_int64()
        self._chk(self.lib.ccsim_dist_poll(self.h, C.byref(done), C.byref(placed)), "ccsim_dist_poll")
        return int(done.value), int(placed.value)

    # ---- ... or the whole sharded run inside the library, over its own RCCL communicator ----
    def dist_comm_init(self, unique_id: bytes, n_ranks: int, rank: int):
        buf = (C.c_uint8 * DIST_ID_BYTES).from_buffer_copy(unique_id)
        self._chk(self.lib.ccsim_dist_comm_init(self.h, buf, int(n_ranks), int(rank)), "ccsim_dist_comm_init")

    # ---- windows of placements on shards (one template, zone spread + hostname anti-affinity): include/ccsim.h ccsim_dist_cw_*
    def dist_cw_eligible(self) -> bool:
        return bool(self.lib.ccsim_dist_cw_eligible(self.h))

    def dist_cw_enable(self, all_ok: bool):
        self._chk(self.lib.ccsim_dist_cw_enable(self.h, 1 if all_ok else 0), "ccsim_dist_cw_enable")

    def dist_cw_buffers(self):
        s, r, b = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._chk(self.lib.ccsim_dist_cw_buffers(self.h, C.byref(s), C.byref(r), C.byref(b)), "ccsim_dist_cw_buffers")
        return int(s.value), int(r.value), int(b.value)

    def dist_cw_scan(self):
        self._chk(self.lib.ccsim_dist_cw_scan(self.h), "ccsim_dist_cw_scan")

    def dist_cw_decide(self):
        self._chk(self.lib.ccsim_dist_cw_decide(self.h), "ccsim_dist_cw_decide")

    def dist_comm_size(self):
        """(ranks, this rank) as the communicator reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int32(), C.c_int32()
        self._chk(self.lib.ccsim_dist_comm_size(self.h, C.byref(n), C.byref(r)), "ccsim_dist_comm_size")
        return int(n.value), int(r.value)

    def dist_sync_tables(self):
        self._chk(self.lib.ccsim_dist_sync_tables(self.h), "ccsim_dist_sync_tables")

    def dist_run(self, max_limit: int = 0, mode: str = "sequential", want_log: bool = False, log_cap: int = 0,
                 reuse_buffers: bool = False) -> M.RunResult:
        """`reuse_buffers`: as in run() -- the shard's per-node counts land in the engine's page-locked result array."""
        rep, per_node, log, ht = self._report(want_log, log_cap, reuse_buffers)
        self._chk(self.lib.ccsim_dist_run(self.h, int(max_limit), MODES[mode], C.byref(rep)), "ccsim_dist_run")
        res = self._result(rep, per_node, log, ht, reuse_buffers)
        if reuse_buffers:
            import weakref

            self._pin_results = [r for r in getattr(self, "_pin_results", []) if r() is not None] + [weakref.ref(res)]
        return res

    # ---- ... or the persistent level kernel across the GPUs: mailbox form (include/ccsim.h ccsim_dist_mbox_*) ----
    def dist_mbox_info(self) -> bytes:
        buf = (C.c_uint8 * MBOX_INFO_BYTES)()
        self._chk(self.lib.ccsim_dist_mbox_info(self.h, buf), "ccsim_dist_mbox_info")
        return bytes(buf)

    def dist_mbox_connect(self, infos, rank: int):
        """`infos`: every rank's dist_mbox_info(), in rank order."""
        blob = b"".join(infos)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._chk(self.lib.ccsim_dist_mbox_connect(self.h, buf, len(infos), int(rank)), "ccsim_dist_mbox_connect")

    def dist_mbox_eligible(self) -> bool:
        return bool(self.lib.ccsim_dist_mbox_eligible(self.h))

    def dist_mbox_launch(self):
        self._chk(self.lib.ccsim_dist_mbox_launch(self.h), "ccsim_dist_mbox_launch")

    def dist_mbox_status(self) -> bool:
        ok = C.c_int32()
        self._chk(self.lib.ccsim_dist_mbox_status(self.h, C.byref(ok)), "ccsim_dist_mbox_status")
        return bool(ok.value)

    def dist_mbox_finish(self, all_ok: bool) -> bool:
        """True: the result stands (dist_finish reports it).  False: fall back to the pass protocol from the untouched state."""
        rc = self.lib.ccsim_dist_mbox_finish(self.h, 1 if all_ok else 0)
        if rc < 0:
            self._chk(rc, "ccsim_dist_mbox_finish")
        return rc == 0

    def dist_finish(self, want_log: bool = False, log_cap: int = 0) -> M.RunResult:
        rep, per_node, log, ht = self._report(want_log, log_cap)
        self._chk(self.lib.ccsim_dist_finish(self.h, C.byref(rep)), "ccsim_dist_finish")
        return self._result(rep, per_node, log, ht)
