"""Interned (integer-world) containers handed to the C-ABI: node SoA columns, pod-spec constants
and lookup tables, and the scheduler profile.

These are plain numpy holders -- plumbing between Python callers (tests, bench.py) and
``include/ccsim.h``.  Column order everywhere: 0 = cpu (milli), 1 = memory, 2 = ephemeral-storage,
3+k = scalar/extended resource k  (reference: vendor/k8s.io/kubernetes/pkg/scheduler/framework/types.go:940-950).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

MAX_SCALAR = 8
MAX_RES = 3 + MAX_SCALAR
MAX_TSC = 8
MAX_IPA_KEYS = 4
MAX_IPA_TERMS = 8

# filter plugin bits, default order (apis/config/v1/default_plugins.go:30-58)
F_UNSCHEDULABLE = 1 << 0
F_NODENAME = 1 << 1
F_TAINT = 1 << 2
F_NODEAFFINITY = 1 << 3
F_FIT = 1 << 4
F_TOPOLOGYSPREAD = 1 << 5
F_INTERPODAFFINITY = 1 << 6
F_NODEPORTS = 1 << 7  # runs between NodeAffinity and NodeResourcesFit (the bit order is not the plugin order)
F_ALL = (F_UNSCHEDULABLE | F_NODENAME | F_TAINT | F_NODEAFFINITY | F_FIT | F_TOPOLOGYSPREAD | F_INTERPODAFFINITY
         | F_NODEPORTS)

# reason slots of the terminal histogram
R_UNSCHEDULABLE = 0
R_NODENAME = 1
R_NODEAFFINITY = 2
R_TOO_MANY_PODS = 3
R_RES0 = 4
R_PTS_MISSING_LABEL = R_RES0 + MAX_RES
R_PTS_SKEW = R_PTS_MISSING_LABEL + 1
R_IPA_AFFINITY = R_PTS_SKEW + 1
R_IPA_ANTI = R_IPA_AFFINITY + 1
R_IPA_EXISTING_ANTI = R_IPA_ANTI + 1
R_NODEPORTS = R_IPA_EXISTING_ANTI + 1
# the volume plugins in filter order: slot = R_VOL0 + (PodSpec.volume_veto code - 1)  (include/ccsim.h CCSIM_R_VOL*)
VOL_DISK_CONFLICT, VOL_RWOP, VOL_MAX_COUNT, VOL_NODE_AFFINITY, VOL_NO_PV, VOL_PV_NOT_EXIST, VOL_ZONE = 1, 2, 3, 4, 5, 6, 7
VOL_CODES = 7
VOL_LAST_UNSCHEDULABLE = 3  # codes 1..3 are plain Unschedulable, 4..7 UnschedulableAndUnresolvable
R_VOL0 = R_NODEPORTS + 1
NREASON = R_VOL0 + VOL_CODES

STOP_UNSCHEDULABLE = 0
STOP_LIMIT = 1
STOP_NO_NODES = 2


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


@dataclass
class NodesSoA:
    """Structure-of-arrays snapshot of NodeInfo (types.go:160-200) in canonical node order
    (backend/cache/node_tree.go:119-143)."""

    alloc: List[np.ndarray]  # [3 + n_scalar] int64[n]
    alloc_pods: np.ndarray  # int32[n]
    req: List[np.ndarray]  # [3 + n_scalar] int64[n]   Requested
    nz_mcpu: np.ndarray  # int64[n]  NonZeroRequested.MilliCPU
    nz_mem: np.ndarray  # int64[n]
    pod_count: np.ndarray  # int32[n]  len(NodeInfo.Pods)
    taintset_id: np.ndarray  # int32[n]
    unschedulable: np.ndarray  # uint8[n]
    label_cols: List[np.ndarray] = field(default_factory=list)  # int32[n] value ids, 0 = absent
    # optional string side (never crosses the ABI)
    names: Optional[List[str]] = None
    scalar_names: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.alloc = [_i64(a) for a in self.alloc]
        self.req = [_i64(a) for a in self.req]
        self.alloc_pods = _i32(self.alloc_pods)
        self.nz_mcpu = _i64(self.nz_mcpu)
        self.nz_mem = _i64(self.nz_mem)
        self.pod_count = _i32(self.pod_count)
        self.taintset_id = _i32(self.taintset_id)
        self.unschedulable = _u8(self.unschedulable)
        self.label_cols = [_i32(c) for c in self.label_cols]
        assert len(self.alloc) == len(self.req) and len(self.alloc) >= 3

    @property
    def n(self) -> int:
        return int(self.alloc_pods.shape[0])

    @property
    def n_scalar(self) -> int:
        return len(self.alloc) - 3

    def copy(self) -> "NodesSoA":
        return NodesSoA(
            alloc=[a.copy() for a in self.alloc],
            alloc_pods=self.alloc_pods.copy(),
            req=[a.copy() for a in self.req],
            nz_mcpu=self.nz_mcpu.copy(),
            nz_mem=self.nz_mem.copy(),
            pod_count=self.pod_count.copy(),
            taintset_id=self.taintset_id.copy(),
            unschedulable=self.unschedulable.copy(),
            label_cols=[c.copy() for c in self.label_cols],
            names=self.names,
            scalar_names=list(self.scalar_names),
        )

    def slice(self, lo: int, hi: int) -> "NodesSoA":
        """Contiguous node-range shard [lo, hi) (multi-GPU partitioning)."""
        s = slice(lo, hi)
        return NodesSoA(
            alloc=[a[s] for a in self.alloc],
            alloc_pods=self.alloc_pods[s],
            req=[a[s] for a in self.req],
            nz_mcpu=self.nz_mcpu[s],
            nz_mem=self.nz_mem[s],
            pod_count=self.pod_count[s],
            taintset_id=self.taintset_id[s],
            unschedulable=self.unschedulable[s],
            label_cols=[c[s] for c in self.label_cols],
            names=self.names[lo:hi] if self.names else None,
            scalar_names=list(self.scalar_names),
        )


# A requirement is (label column index, uint8 table over the column's value ids incl. 0 = absent)
Requirement = Tuple[int, np.ndarray]


@dataclass
class SpreadConstraint:
    """One topologySpreadConstraint after interning (podtopologyspread/common.go:42-56)."""

    col: int
    max_skew: int
    min_domains: int = 1
    hard: bool = True  # DoNotSchedule
    self_match: bool = True
    is_hostname: bool = False
    n_domains: int = 0
    node_match_count: Optional[np.ndarray] = None  # int32[n] existing matching pods per node
    node_included: Optional[np.ndarray] = None  # uint8[n] node inclusion policies
    # engine form of requireAllTopologies = false (see relax_soft): this value id of `col` stands for "the node lacks the key" --
    # counted as a domain like the reference's "" value, scores nothing for this constraint.  0 = none.
    missing_value: int = 0


@dataclass
class InterPodAffinity:
    """InterPodAffinity after interning (interpodaffinity/filtering.go:204-432, scoring.go:81-290).  The
    reference keys its maps by topology pair, so everything is per distinct topology KEY; `*_existing` are
    per-node counts over the snapshot's pods (selectors/namespaces evaluated by the caller), `self`/`*_self`
    describe what one simulated clone (identical to the incoming pod) adds."""

    key_cols: List[int]
    key_ndom: List[int]
    aff_keys: List[int] = field(default_factory=list)  # required affinity terms -> key index
    self_aff: bool = False  # the pod matches all of its own affinity terms
    aff_existing: Optional[np.ndarray] = None  # int32[n] existing pods matching ALL affinity terms
    anti_keys: List[int] = field(default_factory=list)  # required anti-affinity terms -> key index
    anti_self: List[bool] = field(default_factory=list)
    anti_existing: List[Optional[np.ndarray]] = field(default_factory=list)  # per term int32[n]
    exist_anti: List[Optional[np.ndarray]] = field(default_factory=list)  # per key int32[n]
    score_existing: List[Optional[np.ndarray]] = field(default_factory=list)  # per key int64[n]
    score_self: List[int] = field(default_factory=list)  # per key
    entries_existing: int = 0
    self_entries: List[int] = field(default_factory=list)  # per key


@dataclass
class PreemptionSide:
    """What the DefaultPreemption dry run of the terminal cycle needs (HOST ONLY: it never crosses the C ABI; the dry run
    decides the "preemption: ..." part of the FitError message, not a placement).  A victim is an existing pod whose
    priority is lower than the template's (defaultpreemption/default_preemption.go:392-396); clones share the template's
    priority and are never victims."""

    priority: int = 0
    never: bool = False  # spec.preemptionPolicy == Never (default_preemption.go:355-357)
    victim_count: Optional[np.ndarray] = None  # int32[n]; None = no node holds a victim
    victim_req: List[np.ndarray] = field(default_factory=list)  # per resource column int64[n]: what the victims request
    # NodePorts with the victims gone: does a REMAINING existing pod of the node hold a conflicting host port (uint8[n])
    ports_conflict_rest: Optional[np.ndarray] = None
    # uint8[n]: a victim of the node takes part in the PreFilter state of one of the template's topology-coupled FILTERS (matches a
    # hard spread selector / a required (anti)affinity term, or carries an anti-affinity term that matches the template): removing
    # it would change that state (RunPreFilterExtensionRemovePod) -- such nodes are not modelled by the dry run.  None = no such node
    victim_interacts: Optional[np.ndarray] = None
    # PodSpec.volume_veto with the node's victims gone (uint8[n], volumes.veto_with_victims_gone); None = no node is rejected then
    volume_veto_rest: Optional[np.ndarray] = None


@dataclass
class PodSpec:
    """Pod-side constants precomputed on the host."""

    req: np.ndarray  # int64[3 + n_scalar]
    nz_mcpu: int
    nz_mem: int
    has_scalar_entries: bool = False
    taint_filter_ok: np.ndarray = field(default_factory=lambda: np.ones(1, np.uint8))
    taint_prefer_cnt: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    tolerates_unschedulable: bool = False
    affinity_filter_active: bool = False
    has_node_selector: bool = False
    node_selector: List[Requirement] = field(default_factory=list)
    has_required_terms: bool = False
    required: List[List[Requirement]] = field(default_factory=list)
    preferred: List[Tuple[int, List[Requirement]]] = field(default_factory=list)  # (weight, term)
    spread: List[SpreadConstraint] = field(default_factory=list)
    ipa: Optional[InterPodAffinity] = None
    # NodePorts (plugins/nodeports/node_ports.go:67-176): the pod asks for host ports; which nodes' EXISTING pods
    # already hold a conflicting one (uint8[n]; clones conflict with one another by construction)
    has_host_ports: bool = False
    host_ports_conflict: Optional[np.ndarray] = None
    # ImageLocality (plugins/imagelocality/image_locality.go:54-115): per-node score 0..100, uint8[n]; None = 0
    image_score: Optional[np.ndarray] = None
    # VolumeRestrictions / NodeVolumeLimits / VolumeBinding / VolumeZone (after NodeResourcesFit, default_plugins.go:41-44): uint8[n], the
    # code VOL_* of the first of them that rejects the node against the snapshot's pods (None = none); volume_exclusive: the pod's
    # disks conflict with a clone's (volume_restrictions.go:105-150), a node takes at most one clone
    volume_exclusive: bool = False
    volume_veto: Optional[np.ndarray] = None
    # host only (volumes.py): a volume plugin's PreFilter rejects the pod on every node (its message); a ReadWriteOncePod claim nobody
    # uses yet (the first clone takes it: the hosts then set the pod again with VOL_RWOP on every node)
    prefilter_reject: Optional[str] = None
    rwop_capacity_one: bool = False
    preempt: Optional[PreemptionSide] = None  # host only, see PreemptionSide
    # PodTopologySpread scores with requireAllTopologies = false (scoring.go:140): the pod has no constraints of its own and
    # `spread` holds the plugin's system defaults (plugin.go:48-59).  The oracle takes this form literally (label id 0 = key missing);
    # the engine takes the form relax_soft() derives.
    soft_relaxed: bool = False

    def __post_init__(self):
        self.req = _i64(self.req)
        self.taint_filter_ok = _u8(self.taint_filter_ok)
        self.taint_prefer_cnt = _i32(self.taint_prefer_cnt)
        if self.host_ports_conflict is not None:
            self.host_ports_conflict = _u8(self.host_ports_conflict)
        if self.image_score is not None:
            self.image_score = _u8(self.image_score)
        if self.volume_veto is not None:
            self.volume_veto = _u8(self.volume_veto)


def relax_soft(nodes: "NodesSoA", pod: "PodSpec"):
    """The engine form of a pod whose soft spread constraints are scored with requireAllTopologies = false
    (podtopologyspread/scoring.go:61-115,140,147-178,205-219): no node is ignored, a missing key is the value "" when the
    candidate domains are sized and counted, and a node scores nothing for a constraint whose key it lacks.  Per constraint
    whose column has nodes without the key a NEW label column is appended in which those nodes carry one more value id
    (n_domains + 1), named by `missing_value`: every node then "has" every key -- nobody is ignored, the extra id is sized and
    counted like any domain -- and the engine skips the credit where it meets that id.  Returns (nodes, pod) copies; the
    originals (what the oracle takes) are untouched.  Must be applied to the WHOLE snapshot before it is sharded."""
    import copy

    if not pod.soft_relaxed:
        return nodes, pod
    nodes2, pod2 = copy.copy(nodes), copy.copy(pod)
    nodes2.label_cols = list(nodes.label_cols)
    pod2.spread = [copy.copy(k) for k in pod.spread]
    pod2.soft_relaxed = False
    for k in pod2.spread:
        if k.hard:
            continue
        col = np.asarray(nodes2.label_cols[k.col])
        if not (col == 0).any():
            continue
        if k.is_hostname:
            raise NotImplementedError("system default spreading on a cluster with a node that lacks kubernetes.io/hostname")
        nodes2.label_cols.append(np.where(col == 0, k.n_domains + 1, col).astype(np.int32))
        k.col, k.n_domains, k.missing_value = len(nodes2.label_cols) - 1, k.n_domains + 1, k.n_domains + 1
    return nodes2, pod2


@dataclass
class Profile:
    """The scheduler profile: enabled filters, score weights, plugin args
    (apis/config/v1/default_plugins.go:30-58, defaults.go:33-36,229-245)."""

    filter_mask: int = F_ALL
    w_taint: int = 3
    w_nodeaffinity: int = 2
    w_fit: int = 1
    w_balanced: int = 1
    w_topologyspread: int = 2
    w_interpodaffinity: int = 2
    fit_res: Tuple[int, ...] = (0, 1)
    fit_res_w: Tuple[int, ...] = (1, 1)
    bal_res: Tuple[int, ...] = (0, 1)
    percentage_of_nodes_to_score: int = 100  # 0 = adaptive (schedule_one.go:697-723)
    w_imagelocality: int = 1

    @staticmethod
    def default() -> "Profile":
        return Profile()

    @staticmethod
    def fit_only() -> "Profile":
        """BASELINE config 2: NodeResourcesFit Filter + LeastAllocated Score only."""
        return Profile(filter_mask=F_FIT, w_taint=0, w_nodeaffinity=0, w_fit=1, w_balanced=0, w_topologyspread=0,
                       w_interpodaffinity=0, w_imagelocality=0)


@dataclass
class RunResult:
    placed: int
    stop: int
    per_node_count: np.ndarray
    log: Optional[np.ndarray]
    hist: np.ndarray  # int64[NREASON]
    hist_taintset: np.ndarray  # int64[n_taintsets]
    n_code_unschedulable: int
    rounds: int = 0
    evaluated_total: int = 0
    last_evaluated: int = 0
    last_feasible: int = 0
    scans: int = 0
    kernel_ns: int = 0
    pass_kernel_ns: int = 0
    pass_launches: int = 0
    bytes_per_scan: int = 0
    per_spec_count: Optional[np.ndarray] = None  # several pod specs: placements per spec
    prefilter_msg: Optional[str] = None          # host only: the terminal cycle was rejected by a PreFilter plugin (FitError.Diagnosis.PreFilterMsg)
    stop_spec: int = -1                          # ... and the spec whose pod was Unschedulable
