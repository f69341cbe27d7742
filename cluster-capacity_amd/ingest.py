"""Snapshot ingest: Kubernetes objects (parsed YAML/JSON dicts: Nodes, Pods, the pod spec to simulate) -> the
interned integer world of model.py.  Host-side mirror of what the reference does with strings before and inside
the scheduling loop; nothing here is on the hot path.

Reference (paths relative to the reference root; S/ = vendor/k8s.io/kubernetes/pkg/scheduler):
  which objects are copied          pkg/framework/simulator.go:176-295 (non-terminal pods, nodes minus --exclude-nodes)
  Quantity.Value / MilliValue       vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go:813-834 (ceil)
  pod requests                      vendor/k8s.io/component-helpers/resource/helpers.go:144-251 (sum containers, max init, + overhead)
  NonZero requests                  S/framework/types.go:1095-1124 (100m / 200Mi per container without the request)
  NodeInfo.AddPod                   S/framework/types.go:345-350,409-428
  node order                        S/backend/cache/node_tree.go:119-143, component-helpers/node/topology/helpers.go:31-58
  taints / tolerations              component-helpers/scheduling/corev1/helpers.go:63-101, api/core/v1/toleration.go:38-57
  node selector requirements        apimachinery/pkg/labels/selector.go:246-293, component-helpers/.../nodeaffinity.go
  label selectors                   apimachinery/pkg/apis/meta/v1/helpers.go:36-75
  spread constraints                S/framework/plugins/podtopologyspread/common.go:42-159
  inter-pod affinity terms          vendor/k8s.io/kube-scheduler/framework/types.go:379-384, S/framework/types.go:927-935
  host ports                        S/util/utils.go:175-210 GetHostPorts, kube-scheduler/framework/types.go:455-538 HostPortInfo
  image states                      S/backend/cache/cache.go:680-703, S/framework/plugins/imagelocality/image_locality.go:54-127
"""
from __future__ import annotations

import math
import re
from fractions import Fraction
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import model as M

DEFAULT_MILLI_CPU = 100  # S/util/pod_resources.go:28-31
DEFAULT_MEMORY = 200 * 1024 * 1024
HOSTNAME = "kubernetes.io/hostname"
UNSCHED_TAINT = "node.kubernetes.io/unschedulable"

_BIN = {"Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40, "Pi": 2**50, "Ei": 2**60}
_DEC = {"n": Fraction(1, 10**9), "u": Fraction(1, 10**6), "m": Fraction(1, 1000), "": 1, "k": 10**3, "M": 10**6,
        "G": 10**9, "T": 10**12, "P": 10**15, "E": 10**18}
_Q = re.compile(r"^([+-]?[0-9]*\.?[0-9]*)(?:([eE][+-]?[0-9]+)|(Ki|Mi|Gi|Ti|Pi|Ei|n|u|m|k|M|G|T|P|E)?)$")


def parse_quantity(q) -> Fraction:
    """resource.Quantity as an exact rational."""
    if isinstance(q, (int, float)):
        return Fraction(str(q))
    m = _Q.match(str(q).strip())
    if not m or m.group(1) in ("", "+", "-", "."):
        raise ValueError(f"bad quantity {q!r}")
    num = Fraction(m.group(1))
    if m.group(2):
        return num * Fraction(10) ** int(m.group(2)[1:])
    suf = m.group(3) or ""
    return num * (_BIN[suf] if suf in _BIN else _DEC[suf])


def quantity_format(q) -> str:
    """The Format a parsed Quantity carries (quantity.go:283-384 ParseQuantity -> suffix.go interpret): binary suffix -> BinarySI,
    e / E exponent -> DecimalExponent, anything else -> DecimalSI."""
    m = _Q.match(str(q).strip())
    if m and m.group(2):
        return "DecimalExponent"
    return "BinarySI" if m and (m.group(3) or "") in _BIN else "DecimalSI"


_DEC_SUFFIX = {-9: "n", -6: "u", -3: "m", 0: "", 3: "k", 6: "M", 9: "G", 12: "T", 15: "P", 18: "E"}


def quantity_canonical(v: Fraction, fmt: str) -> str:
    """Quantity.String() for a non-negative value (quantity.go:424-461 CanonicalizeBytes; amount.go:257-293; math.go:262-287):
    BinarySI prints an integer >= 1024 as <n><Ki|Mi|..> with every factor of 1024 removed, and falls back to DecimalSI below 1024
    or for a fractional value; the decimal forms print mantissa x 10^exponent with the trailing zeros removed and the exponent
    lowered to a multiple of 3 (12000 -> 12k, 1200 -> 1200, 1.5 -> 1500m).  Values are kept to the nano, rounded up, as
    ParseQuantity does."""
    n = math.ceil(v * 10**9)  # nano units
    if n == 0:
        return "0"
    if fmt == "BinarySI":
        if n < 1024 * 10**9 or n % 10**9:
            fmt = "DecimalSI"
        else:
            m, t = n // 10**9, 0
            while m >= 1024 and m % 1024 == 0:
                m, t = m // 1024, t + 1
            return f"{m}{['', 'Ki', 'Mi', 'Gi', 'Ti', 'Pi', 'Ei'][t]}"
    m, e = n, -9
    while m >= 10 and m % 10 == 0:
        m, e = m // 10, e + 1
    if e % 3 == 1:  # (Go's remainder keeps the sign of the dividend: 1 and -2 are Python's 1; 2 and -1 are Python's 2)
        m, e = m * 10, e - 1
    elif e % 3 == 2:
        m, e = m * 100, e - 2
    if fmt == "DecimalExponent":
        return f"{m}e{e}" if e else f"{m}"
    return f"{m}{_DEC_SUFFIX.get(e, '')}"


def _in_range(v: int, q) -> int:
    """The snapshot's integers: a single quantity is refused beyond 2^60 (1Ei -- no node holds that); sums that leave int64 are refused
    when they are stored (numpy raises OverflowError): never a silent wrap-around."""
    if abs(v) > 1 << 60:
        raise ValueError(f"quantity '{str(q).strip()}' is out of range (beyond 2^60)")
    return v


def value(q) -> int:
    """Quantity.Value(): rounded up to an integer (quantity.go:813-820)."""
    return _in_range(math.ceil(parse_quantity(q)), q)


def milli_value(q) -> int:
    """Quantity.MilliValue(): rounded up to an integer number of thousandths (quantity.go:822-834)."""
    return _in_range(math.ceil(parse_quantity(q) * 1000), q)


def _res(rl: Optional[dict], name: str) -> int:
    if not rl or name not in rl:
        return 0
    return milli_value(rl[name]) if name == "cpu" else value(rl[name])


_QNAME = re.compile(r"([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9]")  # apimachinery/pkg/util/validation/validation.go:29-35
_DNS1123_LABEL = r"[a-z0-9]([-a-z0-9]*[a-z0-9])?"                   # :176
_DNS1123_SUBDOMAIN = re.compile(_DNS1123_LABEL + r"(\." + _DNS1123_LABEL + r")*")  # :205


def is_qualified_name(value: str) -> bool:
    """validation.IsQualifiedName (apimachinery/pkg/util/validation/validation.go:41-70): [dns-subdomain "/"] name, name <= 63 chars."""
    parts = value.split("/")
    if len(parts) == 1:
        name = parts[0]
    elif len(parts) == 2:
        prefix, name = parts
        if not prefix or len(prefix) > 253 or not _DNS1123_SUBDOMAIN.fullmatch(prefix):
            return False
    else:
        return False
    return 0 < len(name) <= 63 and _QNAME.fullmatch(name) is not None


def is_scalar_resource(name: str) -> bool:
    """schedutil.IsScalarResourceName (S/util/utils.go:140-143) = extended || hugepages-* || *kubernetes.io/* || attachable-volumes-*
    (pkg/apis/core/v1/helper/helpers.go:36-66,133-135).  Anything else that is not cpu / memory / ephemeral-storage is DROPPED by
    the scheduler's Resource.Add (S/framework/types.go), e.g. an unqualified "foo" or a "requests."-prefixed name."""
    prefixed_native = "kubernetes.io/" in name
    native = "/" not in name or prefixed_native
    extended = not native and not name.startswith("requests.") and is_qualified_name("requests." + name)
    return extended or name.startswith("hugepages-") or prefixed_native or name.startswith("attachable-volumes-")


def _pod_level_supported(name: str) -> bool:
    """IsSupportedPodLevelResource (component-helpers/resource/helpers.go): cpu, memory, hugepages-*."""
    return name in ("cpu", "memory") or name.startswith("hugepages-")


def _aggregate(spec: dict, name: str, default: Optional[int] = None) -> int:
    """PodRequests for one resource (helpers.go:144-251): sum of the containers; restartable (sidecar) init containers add to
    the sum; InitContainerUse(i) = the i-th init container + the sidecars before it, and the pod needs at least the largest
    of those; pod-level requests (spec.resources) override the aggregate for the resources they may carry; + overhead.
    `default` stands in for a container that does not name the resource (NonMissingContainerRequests, types.go:1095-1124)."""
    def creq(c):
        r = (c.get("resources") or {}).get("requests") or {}
        if name in r:
            return milli_value(r[name]) if name == "cpu" else value(r[name])
        return default if default is not None else 0

    total = sum(creq(c) for c in spec.get("containers") or [])
    restartable = init_max = 0
    for ic in spec.get("initContainers") or []:
        r = creq(ic)
        if ic.get("restartPolicy") == "Always":
            total += r
            restartable += r
            use = restartable
        else:
            use = r + restartable
        init_max = max(init_max, use)
    total = max(total, init_max)
    pod_level = (spec.get("resources") or {}).get("requests") or {}
    if name in pod_level and _pod_level_supported(name):
        total = _res(pod_level, name)
    return total + _res(spec.get("overhead"), name)


def _named_anywhere(spec: dict, name: str) -> bool:
    """Does the pod's aggregated ResourceList hold `name` at all (some container, init container or the pod level names it)?"""
    for c in (spec.get("containers") or []) + (spec.get("initContainers") or []):
        if name in ((c.get("resources") or {}).get("requests") or {}):
            return True
    return name in ((spec.get("resources") or {}).get("requests") or {}) and _pod_level_supported(name)


def _named_anywhere_any(spec: dict, name: str) -> bool:
    """Does the template's own ResourceList hold `name` (container, init container, pod level or overhead)?"""
    for c in (spec.get("containers") or []) + (spec.get("initContainers") or []):
        if name in ((c.get("resources") or {}).get("requests") or {}):
            return True
    return name in ((spec.get("resources") or {}).get("requests") or {}) or name in (spec.get("overhead") or {})


def pod_requests(spec: dict, names: Sequence[str]):
    """-> (requests per name, non-zero cpu, non-zero memory).  helpers.go:144-251 PodRequests + types.go:700-734, 1095-1124:
    without pod-level requests every container lacking cpu / memory counts 100m / 200Mi; WITH pod-level requests
    (spec.resources.requests non-empty) a default is used only for a resource that neither a container nor the pod level names."""
    out = {n: _aggregate(spec, n) for n in names}
    pod_level_set = bool((spec.get("resources") or {}).get("requests"))

    def non_zero(name, dflt):
        return _aggregate(spec, name, dflt if not pod_level_set or not _named_anywhere(spec, name) else None)

    return out, non_zero("cpu", DEFAULT_MILLI_CPU), non_zero("memory", DEFAULT_MEMORY)


def zone_key(labels: dict) -> str:
    zone = labels.get("failure-domain.beta.kubernetes.io/zone", labels.get("topology.kubernetes.io/zone", ""))
    region = labels.get("failure-domain.beta.kubernetes.io/region", labels.get("topology.kubernetes.io/region", ""))
    if not region and not zone:
        return ""
    return region + ":\x00:" + zone


def canonical_node_order(nodes: List[dict]) -> List[dict]:
    """node_tree.go:119-143: nodes arrive sorted by name (the fake tracker lists lexicographically,
    client-go/testing/fixture.go:847-855), zones in first-seen order, then round robin across zones."""
    zones: Dict[str, List[dict]] = {}
    for n in sorted(nodes, key=lambda n: n["metadata"]["name"]):
        zones.setdefault(zone_key(n["metadata"].get("labels") or {}), []).append(n)
    out, i = [], 0
    while len(out) < len(nodes):
        for z in zones.values():
            if i < len(z):
                out.append(z[i])
        i += 1
    return out


# ---- taints / tolerations -------------------------------------------------------------------------------------
def tolerates(tol: dict, taint: dict) -> bool:
    """toleration.go:38-57 ToleratesTaint."""
    if tol.get("effect") and tol["effect"] != taint.get("effect"):
        return False
    if tol.get("key") and tol["key"] != taint.get("key"):
        return False
    op = tol.get("operator") or "Equal"
    if op == "Exists":
        return True
    if op == "Equal":
        return (tol.get("value") or "") == (taint.get("value") or "")
    return False


def taint_verdict(taints: List[dict], tolerations: List[dict]):
    """-> (filter_ok, untolerated PreferNoSchedule count, first untolerated NoSchedule/NoExecute taint or None)."""
    first = None
    for t in taints:
        if t.get("effect") in ("NoSchedule", "NoExecute") and not any(tolerates(x, t) for x in tolerations):
            first = t
            break
    prefer_tols = [x for x in tolerations if not x.get("effect") or x.get("effect") == "PreferNoSchedule"]
    cnt = sum(1 for t in taints if t.get("effect") == "PreferNoSchedule" and not any(tolerates(x, t) for x in prefer_tols))
    return first is None, cnt, first


# ---- host ports (NodePorts) -------------------------------------------------------------------------------------
def host_ports(spec: dict):
    """util.GetHostPorts (S/util/utils.go:175-210): ports with hostPort > 0 of the restartable init containers and of the
    containers -> [(hostIP, protocol, hostPort)], sanitized as HostPortInfo does ("" -> 0.0.0.0 / TCP, types.go:530-538)."""
    out = []
    cs = [c for c in spec.get("initContainers") or [] if c.get("restartPolicy") == "Always"] + list(spec.get("containers") or [])
    for c in cs:
        for p in c.get("ports") or []:
            hp = int32_field(p.get("hostPort"))
            if hp > 0:
                out.append((p.get("hostIP") or "0.0.0.0", p.get("protocol") or "TCP", hp))
    return out


def ports_conflict(want, used) -> bool:
    """fitsPorts (node_ports.go:164-176) over HostPortInfo.CheckConflict (types.go:499-528); `used` = set of sanitized
    (ip, protocol, port) held by the node's pods.  0.0.0.0 conflicts with every ip on the same (protocol, port)."""
    for ip, proto, port in want:
        for uip, uproto, uport in used:
            if (uproto, uport) == (proto, port) and (ip == "0.0.0.0" or uip == "0.0.0.0" or uip == ip):
                return True
    return False


# ---- image locality ---------------------------------------------------------------------------------------------
_MB = 1024 * 1024


def normalized_image_name(name: str) -> str:
    """image_locality.go:122-127: append :latest when the last path component carries no tag."""
    return name + ":latest" if name.rfind(":") <= name.rfind("/") else name


def image_locality_score(sizes_and_spread, total_nodes: int, n_containers: int) -> int:
    """calculatePriority(sumImageScores) (image_locality.go:84-115); `sizes_and_spread` = (Size, NumNodes) of the pod's
    container images present on the node.  scaledImageScore is fp64: int64(float64(Size) * (NumNodes / total))."""
    total = 0
    for size, num_nodes in sizes_and_spread:
        total += int(float(size) * (float(num_nodes) / float(total_nodes)))
    lo, hi = 23 * _MB, 1000 * _MB * n_containers
    total = lo if total < lo else (hi if total > hi else total)
    q = 100 * (total - lo)
    d = hi - lo
    return int(abs(q) // abs(d)) * (1 if (q >= 0) == (d > 0) else -1)  # Go integer division truncates toward zero


def image_scores(nodes: List[dict], spec: dict) -> Optional[np.ndarray]:
    """Per-node ImageLocality score (uint8) or None when none of the pod's images is on any node.  The image states are
    the scheduler cache's (cache.go:680-703): Size = what the FIRST node added (nodes arrive sorted by name) reports for
    that image name, NumNodes = nodes listing the name."""
    by_name = sorted(range(len(nodes)), key=lambda i: nodes[i]["metadata"]["name"])
    size: Dict[str, int] = {}
    holders: Dict[str, set] = {}
    for i in by_name:
        for img in (nodes[i].get("status") or {}).get("images") or []:
            for nm in img.get("names") or []:
                size.setdefault(nm, int(img.get("sizeBytes") or 0))
                holders.setdefault(nm, set()).add(i)
    cs = list(spec.get("initContainers") or []) + list(spec.get("containers") or [])
    wanted = [normalized_image_name(c.get("image") or "") for c in cs]
    if not any(w in holders for w in wanted):
        return None
    out = np.zeros(len(nodes), np.uint8)
    for i in range(len(nodes)):
        present = [(size[w], len(holders[w])) for w in wanted if w in holders and i in holders[w]]
        out[i] = image_locality_score(present, len(nodes), len(cs))
    return out


# ---- selectors --------------------------------------------------------------------------------------------------
def requirement_matches(key_present: bool, val: Optional[str], op: str, values: List[str]) -> bool:
    """labels.Requirement.Matches (selector.go:246-293)."""
    if op == "In":
        return key_present and val in values
    if op == "NotIn":
        return not key_present or val not in values
    if op == "Exists":
        return key_present
    if op == "DoesNotExist":
        return not key_present
    if op in ("Gt", "Lt"):
        if not key_present or len(values) != 1:
            return False
        a, b = go_parse_int(val), go_parse_int(values[0])
        if a is None or b is None:
            return False
        return a > b if op == "Gt" else a < b
    return False


def int32_field(x, default: int = 0) -> int:
    """An int32 field of an object (weights, maxSkew, minDomains, priority, hostPort): None = absent; anything that is not a plain integer
    inside int32 is refused, as the reference's decoder refuses it (and as host/value.hpp as_int32 does)."""
    if x is None:
        return default
    if isinstance(x, bool) or not isinstance(x, int):  # (a quoted "8080" is a string to the reference's typed decoder: refused there, refused here)
        raise ValueError(f"malformed object: expected an integer, found {x!r}")
    v = int(x)
    if not -(1 << 31) <= v < (1 << 31):
        raise ValueError(f"malformed object: {v} does not fit an int32 field")
    return v


def go_parse_int(text) -> Optional[int]:
    """strconv.ParseInt(text, 10, 64) as labels.Requirement.Matches uses it (apimachinery/pkg/labels/selector.go:264-289): an optional
    sign and decimal digits, nothing else (Python's int() would also take "1_0", " 5" or "٣"), inside int64; None = the error case."""
    if not isinstance(text, str) or not re.fullmatch(r"[+-]?[0-9]+", text):
        return None
    v = int(text)
    return v if -(1 << 63) <= v < (1 << 63) else None


def label_selector_matches(sel: Optional[dict], labels: dict) -> bool:
    """metav1.LabelSelectorAsSelector: nil -> Nothing, {} -> Everything (helpers.go:36-75)."""
    if sel is None:
        return False
    for k, v in (sel.get("matchLabels") or {}).items():
        if labels.get(k) != v:
            return False
    for e in sel.get("matchExpressions") or []:
        if not requirement_matches(e["key"] in labels, labels.get(e["key"]), e["operator"], e.get("values") or []):
            return False
    return True


def selector_empty(sel: Optional[dict]) -> bool:
    return sel is not None and not (sel.get("matchLabels") or {}) and not (sel.get("matchExpressions") or [])


class Interner:
    """label key -> column; label value -> id (0 = key absent).  Columns are created on demand."""

    def __init__(self, nodes: List[dict]):
        self.nodes = nodes
        self.cols: Dict[str, int] = {}
        self.values: List[List[str]] = []  # per column: value strings, index = id - 1
        self.arrays: List[np.ndarray] = []

    def col(self, key: str) -> int:
        if key in self.cols:
            return self.cols[key]
        vals: Dict[str, int] = {}
        ids = np.zeros(len(self.nodes), np.int32)
        for i, n in enumerate(self.nodes):
            lab = dict(n["metadata"].get("labels") or {})
            if key == "metadata.name":
                lab = {key: n["metadata"]["name"]}
            if key in lab:
                ids[i] = vals.setdefault(lab[key], len(vals) + 1)
        self.cols[key] = len(self.arrays)
        self.arrays.append(ids)
        self.values.append(list(vals))
        return self.cols[key]

    def table(self, key: str, op: str, values: List[str]) -> "M.Requirement":
        c = self.col(key)
        t = np.zeros(len(self.values[c]) + 1, np.uint8)
        t[0] = requirement_matches(False, None, op, values)
        for i, v in enumerate(self.values[c]):
            t[i + 1] = requirement_matches(True, v, op, values)
        return (c, t)


def _node_selector_term(it: Interner, term: dict) -> List["M.Requirement"]:
    reqs = [it.table(e["key"], e["operator"], e.get("values") or []) for e in term.get("matchExpressions") or []]
    for f in term.get("matchFields") or []:  # only metadata.name with In / NotIn (nodeaffinity.go:260-293)
        reqs.append(it.table("metadata.name", f["operator"], f.get("values") or []))
    return reqs


# ---- the snapshot -----------------------------------------------------------------------------------------------
ZONE = "topology.kubernetes.io/zone"


def default_selector(sim_pod: dict, service_objs: Sequence[dict], owner_objs: Sequence[dict] = ()) -> Optional[dict]:
    """helper.DefaultSelector (P/helper/spread.go:37-116) as a LabelSelector dict, None when it is empty: the merged selectors of the
    Services of the pod's namespace that select it, plus the selector of the pod's controller (a pod spec copied from a live pod
    carries its ownerReferences): a ReplicationController's map, a ReplicaSet's / StatefulSet's label selector."""
    ns = sim_pod["metadata"].get("namespace") or "default"
    labels = sim_pod["metadata"].get("labels") or {}
    merged: Dict[str, str] = {}
    for svc in service_objs:
        if (svc["metadata"].get("namespace") or "default") != ns:
            continue
        sel = (svc.get("spec") or {}).get("selector")
        if sel is not None and all(labels.get(k) == v for k, v in sel.items()):  # a nil selector matches nothing (spread.go:105-108)
            merged.update(sel)
    exprs: List[dict] = []
    for ref in sim_pod["metadata"].get("ownerReferences") or []:
        if not ref.get("controller"):
            continue
        for o in owner_objs:
            if (o.get("kind"), o["metadata"].get("name"), o["metadata"].get("namespace") or "default") != (ref.get("kind"), ref.get("name"), ns):
                continue
            sel = (o.get("spec") or {}).get("selector") or {}
            if ref.get("kind") == "ReplicationController" and (ref.get("apiVersion") or "v1") == "v1":
                merged.update(sel)
            elif ref.get("kind") in ("ReplicaSet", "StatefulSet") and ref.get("apiVersion") == "apps/v1":  # (helper/spread.go:31-32: the exact GroupVersionKind)
                merged.update(sel.get("matchLabels") or {})  # (selector.Add of the requirements: an equality per matchLabels entry)
                exprs += list(sel.get("matchExpressions") or [])
        break  # (GetControllerOf: the first reference marked controller)
    if not merged and not exprs:
        return None
    return {"matchLabels": merged, "matchExpressions": exprs}


def system_default_constraints(sim_pod: dict, service_objs: Sequence[dict], owner_objs: Sequence[dict] = ()) -> List[dict]:
    """PodTopologySpread's SYSTEM DEFAULT constraints for a pod WITHOUT constraints of its own (P/podtopologyspread/plugin.go:48-59,
    common.go:61-74): hostname maxSkew 3 and zone maxSkew 5, ScheduleAnyway, selector = helper.DefaultSelector; [] when that is empty."""
    if (sim_pod.get("spec") or {}).get("topologySpreadConstraints"):
        return []
    sel = default_selector(sim_pod, service_objs, owner_objs)
    if sel is None:
        return []
    return [{"maxSkew": 3, "topologyKey": HOSTNAME, "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": sel},
            {"maxSkew": 5, "topologyKey": ZONE, "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": sel}]


def default_spreading_applies(sim_pod: dict, service_objs: Sequence[dict], owner_objs: Sequence[dict] = ()) -> bool:
    return bool(system_default_constraints(sim_pod, service_objs, owner_objs))


class Snapshot:
    """Everything the engine needs, plus the strings the report needs.  With several templates (`--podspec` repeated: the
    reference's report layer takes pod i as a clone of template i mod P, report.go:146-171) `pod` / `taint_reasons` describe
    the first one and `pods` / `taint_reasons_all` all of them; the node columns (incl. the label columns every template's
    selectors and topology keys touch) are shared."""

    def __init__(self, nodes: M.NodesSoA, pod: M.PodSpec, names: List[str], taint_reasons: List[str], scalar_names: List[str]):
        self.nodes, self.pod, self.names, self.taint_reasons, self.scalar_names = nodes, pod, names, taint_reasons, scalar_names
        self.pods, self.taint_reasons_all = [pod], [taint_reasons]
        self.default_spreading_unmodelled = False  # system default spreading applies but a node lacks the hostname / zone label (or several templates run)


def _term_matches_pod(term: dict, term_owner_ns: str, pod: dict, ns_labels: Optional[Dict[str, dict]] = None) -> bool:
    """AffinityTerm.Matches (S/framework/types.go:927-935): the pod's namespace is in the term's set, or its namespace's
    labels match the term's namespaceSelector; then the label selector decides.  newAffinityTerm (:879-895): no namespaces
    and no namespaceSelector -> the namespace of the pod that owns the term.  A namespace without a Namespace object in the
    snapshot has no labels (as the scheduler's lister would report)."""
    ns_set = list(term.get("namespaces") or [])
    ns_sel = term.get("namespaceSelector")
    if not ns_set and ns_sel is None:
        ns_set = [term_owner_ns]
    pod_ns = pod["metadata"].get("namespace") or "default"
    if pod_ns not in ns_set and not (ns_sel is not None and label_selector_matches(ns_sel, (ns_labels or {}).get(pod_ns, {}))):
        return False
    return label_selector_matches(term.get("labelSelector"), pod["metadata"].get("labels") or {})


def build_snapshot(node_objs: List[dict], pod_objs: List[dict], sim_pod, exclude_nodes: Sequence[str] = (),
                   hard_pod_affinity_weight: int = 1, namespace_objs: Sequence[dict] = (), service_objs: Sequence[dict] = (),
                   owner_objs: Sequence[dict] = (), system_default_spreading: bool = True, pvc_objs: Sequence[dict] = (),
                   class_objs: Sequence[dict] = (), pv_objs: Optional[Sequence[dict]] = None,
                   volume_plugins: Sequence[str] = ("VolumeRestrictions", "NodeVolumeLimits", "VolumeBinding", "VolumeZone"),
                   volume_plugins_partial: bool = False, csinode_objs: Sequence[dict] = (), attachment_objs: Sequence[dict] = (),
                   dra_enabled: bool = True, dra_partial: bool = False) -> Snapshot:
    """SyncWithClient (simulator.go:176-295: namespaces, nodes, pods) + every per-pod-spec precomputation, in integers.
    `sim_pod`: the template, or a list of templates (cycled round-robin by the simulation)."""
    sim_pods = list(sim_pod) if isinstance(sim_pod, (list, tuple)) else [sim_pod]
    ns_labels = {n["metadata"]["name"]: (n["metadata"].get("labels") or {}) for n in namespace_objs}

    nodes = canonical_node_order([n for n in node_objs if n["metadata"]["name"] not in set(exclude_nodes)])
    N = len(nodes)
    names = [n["metadata"]["name"] for n in nodes]
    index = {nm: i for i, nm in enumerate(names)}

    # resources: cpu, memory, ephemeral-storage + every scalar resource a template names
    req_names = set()
    for sp in sim_pods:
        spec = sp.get("spec") or {}
        for c in (spec.get("containers") or []) + (spec.get("initContainers") or []):
            req_names |= set(((c.get("resources") or {}).get("requests") or {}).keys())
        req_names |= set(((spec.get("resources") or {}).get("requests") or {}).keys())  # pod-level requests (hugepages-*)
        req_names |= set((spec.get("overhead") or {}).keys())
    scalars = sorted(n for n in req_names if is_scalar_resource(n))
    if len(scalars) > M.MAX_SCALAR:  # never drop a resource silently: the Fit filter would over-estimate the capacity
        raise ValueError(f"the pod names {len(scalars)} scalar/extended resources; at most {M.MAX_SCALAR} are supported")
    res_names = ["cpu", "memory", "ephemeral-storage"] + scalars

    alloc = [np.zeros(N, np.int64) for _ in res_names]
    alloc_pods = np.zeros(N, np.int32)
    for i, n in enumerate(nodes):
        a = (n.get("status") or {}).get("allocatable") or {}
        for c, r in enumerate(res_names):
            alloc[c][i] = _res(a, r)
        ap = value(a.get("pods", 0))
        if not 0 <= ap <= 2**31 - 1:
            raise ValueError(f"node {(n.get('metadata') or {}).get('name', '')}: allocatable pods {a.get('pods')} is out of range")
        alloc_pods[i] = ap
    req = [np.zeros(N, np.int64) for _ in res_names]
    nzc, nzm, pcount = np.zeros(N, np.int64), np.zeros(N, np.int64), np.zeros(N, np.int32)
    live = []  # non-terminal pods bound to a kept node (simulator.go:193-200)
    for p in pod_objs:
        phase = (p.get("status") or {}).get("phase")
        node = (p.get("spec") or {}).get("nodeName")
        if phase in ("Succeeded", "Failed") or node not in index:
            continue
        live.append(p)
        i = index[node]
        r, c0, m0 = pod_requests(p["spec"], res_names)
        for c, rn in enumerate(res_names):
            req[c][i] = int(req[c][i]) + r[rn]  # (Python integers: a sum beyond int64 raises OverflowError at the store, numpy's += would wrap)
        nzc[i] = int(nzc[i]) + c0
        nzm[i] = int(nzm[i]) + m0
        pcount[i] += 1

    # taints -> distinct taint sets (per node); what a template's tolerations make of each set is the template's business
    sets: Dict[str, int] = {}
    set_taints: List[List[dict]] = []
    ts_id = np.zeros(N, np.int32)
    for i, n in enumerate(nodes):
        taints = (n.get("spec") or {}).get("taints") or []
        key = repr([(t.get("key"), t.get("value"), t.get("effect")) for t in taints])
        if key not in sets:
            sets[key] = len(sets)
            set_taints.append(taints)
        ts_id[i] = sets[key]
    unsched = np.array([1 if (n.get("spec") or {}).get("unschedulable") else 0 for n in nodes], np.uint8)
    it = Interner(nodes)  # shared by the templates: a label column per key any of them touches

    ctx = dict(nodes=nodes, N=N, index=index, live=live, ns_labels=ns_labels, res_names=res_names, scalars=scalars, set_taints=set_taints,
               ts_id=ts_id, it=it, hard_pod_affinity_weight=hard_pod_affinity_weight,
               default_spreading=(service_objs, owner_objs) if system_default_spreading else None, n_templates=len(sim_pods),
               pvc_objs=pvc_objs, class_objs=class_objs, pv_objs=pv_objs, volume_plugins=tuple(volume_plugins),
               volume_plugins_partial=volume_plugins_partial, csinode_objs=csinode_objs, attachment_objs=attachment_objs,
               dra_enabled=dra_enabled, dra_partial=dra_partial)
    sides = [_template_side(dict(ctx, template_index=t), sp) for t, sp in enumerate(sim_pods)]
    soa = M.NodesSoA(alloc=alloc, alloc_pods=alloc_pods, req=req, nz_mcpu=nzc, nz_mem=nzm, pod_count=pcount,
                     taintset_id=ts_id, unschedulable=unsched, label_cols=[a for a in it.arrays] or [], names=names,
                     scalar_names=scalars)
    snap = Snapshot(soa, sides[0][0], names, sides[0][1], scalars)
    snap.default_spreading_unmodelled = bool(ctx.get("default_spreading_unmodelled"))
    snap.pods, snap.taint_reasons_all = [p for p, _ in sides], [r for _, r in sides]
    if len(sides) > 1:
        _check_templates_disjoint(sim_pods)
        _check_template_volumes_disjoint(sim_pods, pvc_objs)
    return snap


def _check_template_volumes_disjoint(sim_pods: List[dict], pvc_objs: Sequence[dict]):
    """Several templates: a clone's disks exclude clones of the SAME template from its node (volume_exclusive); a disk or a
    ReadWriteOncePod claim shared by two templates would exclude the other's clones too, which nothing tracks."""
    from . import volumes as V

    rwop = {((o.get("metadata") or {}).get("namespace") or "default", (o.get("metadata") or {}).get("name", "")) for o in pvc_objs
            if "ReadWriteOncePod" in ((o.get("spec") or {}).get("accessModes") or [])}

    def claims(p):
        ns = (p.get("metadata") or {}).get("namespace") or "default"
        return {(ns, (v["persistentVolumeClaim"] or {}).get("claimName", "")) for v in (p.get("spec") or {}).get("volumes") or []
                if v.get("persistentVolumeClaim") is not None}
    for a_i, a in enumerate(sim_pods):
        for b_i, b in enumerate(sim_pods):
            if a_i >= b_i:
                continue
            if V.pod_conflicts((a.get("spec") or {}).get("volumes") or [], (b.get("spec") or {}).get("volumes") or []):
                raise NotImplementedError(f"templates {a_i} and {b_i} mount the same disk: conflicts between clones of different templates are not modelled")
            if claims(a) & claims(b) & rwop:
                raise NotImplementedError(f"templates {a_i} and {b_i} share a ReadWriteOncePod claim: not modelled")


def _check_templates_disjoint(sim_pods: List[dict]):
    """Several templates: what a clone contributes to the plugin state of later cycles is kept per template (the engine's
    ccsim_set_pods contract, include/ccsim.h): no selector of one template may match the clones of another."""
    for a_i, a in enumerate(sim_pods):
        spec = a.get("spec") or {}
        sels = [c.get("labelSelector") for c in spec.get("topologySpreadConstraints") or []]
        aff = spec.get("affinity") or {}
        for kind in ("podAffinity", "podAntiAffinity"):
            k = aff.get(kind) or {}
            sels += [t.get("labelSelector") for t in k.get("requiredDuringSchedulingIgnoredDuringExecution") or []]
            sels += [(t.get("podAffinityTerm") or {}).get("labelSelector") for t in k.get("preferredDuringSchedulingIgnoredDuringExecution") or []]
        for b_i, b in enumerate(sim_pods):
            if b_i == a_i:
                continue
            for sel in sels:
                if sel is not None and not selector_empty(sel) and label_selector_matches(sel, b["metadata"].get("labels") or {}):
                    raise NotImplementedError(f"several templates: a selector of template {a_i} matches the labels of template {b_i}")


def _template_side(ctx: dict, sim_pod: dict):
    """Everything one template contributes: -> (PodSpec, reason string per taint set)."""
    nodes, N, index, live, ns_labels, res_names, scalars = (ctx[k] for k in ("nodes", "N", "index", "live", "ns_labels", "res_names", "scalars"))
    ts_id, it, hard_pod_affinity_weight = ctx["ts_id"], ctx["it"], ctx["hard_pod_affinity_weight"]

    def tm(term, owner_ns, pod):
        return _term_matches_pod(term, owner_ns, pod, ns_labels)

    spec = sim_pod.get("spec") or {}
    sim_ns = sim_pod["metadata"].get("namespace") or "default"
    sim_labels = sim_pod["metadata"].get("labels") or {}
    preq, nz_cpu, nz_mem = pod_requests(spec, res_names)
    own_scalars = [n for n in scalars if _named_anywhere_any(spec, n)]

    tolerations = spec.get("tolerations") or []
    ok, cnt, reasons = [], [], []
    for taints in ctx["set_taints"]:
        f, c, first = taint_verdict(taints, tolerations)
        ok.append(f)
        cnt.append(c)
        # taint_toleration.go:119
        reasons.append("" if first is None else f"node(s) had untolerated taint {{{first.get('key')}: {first.get('value') or ''}}}")
    tol_unsched = any(tolerates(t, {"key": UNSCHED_TAINT, "effect": "NoSchedule"}) for t in tolerations)

    # node affinity / node selector
    aff = (spec.get("affinity") or {}).get("nodeAffinity") or {}
    node_selector = spec.get("nodeSelector")
    required = (aff.get("requiredDuringSchedulingIgnoredDuringExecution") or {}).get("nodeSelectorTerms")
    sel_reqs = [it.table(k, "In", [v]) for k, v in (node_selector or {}).items()]
    req_terms = [_node_selector_term(it, t) for t in (required or [])]
    pref = [(int32_field(t.get("weight")), _node_selector_term(it, t["preference"]))
            for t in aff.get("preferredDuringSchedulingIgnoredDuringExecution") or []]
    affinity_active = bool(node_selector) or required is not None

    def node_matches_required(i: int) -> bool:  # RequiredNodeAffinity.Match, for the spread inclusion policy
        def term_ok(reqs, empty):
            return empty if not reqs else all(t[it.arrays[c][i]] for c, t in reqs)
        if node_selector and not term_ok(sel_reqs, True):
            return False
        if required is not None and not any(term_ok(t, False) for t in req_terms):
            return False
        return True

    pod = M.PodSpec(
        req=np.array([preq[r] for r in res_names], np.int64), nz_mcpu=nz_cpu, nz_mem=nz_mem,
        has_scalar_entries=bool(own_scalars), taint_filter_ok=np.array(ok, np.uint8), taint_prefer_cnt=np.array(cnt, np.int32),
        tolerates_unschedulable=tol_unsched, affinity_filter_active=affinity_active,
        has_node_selector=node_selector is not None and len(node_selector) > 0, node_selector=sel_reqs,
        has_required_terms=required is not None, required=req_terms, preferred=pref)

    # NodePorts: which nodes' existing pods already hold one of the pod's host ports
    want = host_ports(spec)
    if want:
        pod.has_host_ports = True
        used: Dict[int, set] = {}
        for p in live:
            hp = host_ports(p["spec"])
            if hp:
                used.setdefault(index[p["spec"]["nodeName"]], set()).update(hp)
        conflict = np.array([1 if i in used and ports_conflict(want, used[i]) else 0 for i in range(N)], np.uint8)
        pod.host_ports_conflict = conflict if conflict.any() else None
    # ImageLocality: per-node score from node.status.images
    pod.image_score = image_scores(nodes, spec)
    # DefaultPreemption dry run (report only): what removing every lower-priority pod of a node would free
    # (corev1helpers.PodPriority: spec.priority, 0 when unset; default_preemption.go:392-396)
    prio = int32_field(spec.get("priority"))
    pre = M.PreemptionSide(priority=prio, never=spec.get("preemptionPolicy") == "Never")
    victims = [p for p in live if int32_field(p["spec"].get("priority")) < prio]
    victim_ids = {id(p) for p in victims}
    # nodes with a victim whose removal would change the PreFilter state of a topology-coupled FILTER of this template (it
    # matches a hard spread selector or a required (anti)affinity term, or carries an anti-affinity term matching the template)
    interacts = np.zeros(N, np.uint8)
    if victims:
        pre.victim_count = np.zeros(N, np.int32)
        pre.victim_req = [np.zeros(N, np.int64) for _ in res_names]
        for p in victims:
            i = index[p["spec"]["nodeName"]]
            r, _, _ = pod_requests(p["spec"], res_names)
            pre.victim_count[i] += 1
            for c, rn in enumerate(res_names):
                pre.victim_req[c][i] += r[rn]
        if want:
            vic = set(map(id, victims))
            rest: Dict[int, set] = {}
            for p in live:
                hp = host_ports(p["spec"]) if id(p) not in vic else None
                if hp:
                    rest.setdefault(index[p["spec"]["nodeName"]], set()).update(hp)
            pre.ports_conflict_rest = np.array([1 if i in rest and ports_conflict(want, rest[i]) else 0 for i in range(N)], np.uint8)
    pod.preempt = pre
    # VolumeRestrictions / NodeVolumeLimits / VolumeBinding / VolumeZone: the object side (volumes.py) -> per-node verdict codes, the
    # clones' own disks, the PreFilter rejections.  (inline csi volumes only count against CSINode limits, which the simulated cluster
    # does not have: nodevolumelimits/csi.go:265-290.)  DynamicResources stays refused.
    from . import volumes as V
    if spec.get("volumes") and ctx.get("volume_plugins_partial"):
        raise NotImplementedError("the scheduler configuration disables only the filter point of a volume plugin: a pod with volumes is not modelled under it")
    vs = V.volume_side(sim_pod, nodes, live, index, ctx.get("pvc_objs") or (), ctx.get("class_objs") or (), ctx.get("pv_objs"),
                       ctx.get("volume_plugins") or V.PLUGINS, ctx.get("csinode_objs") or (), ctx.get("attachment_objs") or (),
                       clone_index=ctx.get("template_index", 0))
    pod.volume_veto, pod.volume_exclusive = vs.veto, vs.exclusive
    pod.prefilter_reject, pod.rwop_capacity_one = vs.prefilter_reject, vs.rwop_capacity_one
    if victims and vs.veto is not None and vs.prefilter_reject is None:  # (DefaultPreemption's dry run: the verdicts once a node's victims are gone)
        pre.volume_veto_rest = V.veto_with_victims_gone(sim_pod, nodes, live, victims, index, vs, pvc_objs=ctx.get("pvc_objs") or (), class_objs=ctx.get("class_objs") or (),
                                                        pv_objs=ctx.get("pv_objs"), enabled=ctx.get("volume_plugins") or V.PLUGINS, csinode_objs=ctx.get("csinode_objs") or (),
                                                        attachment_objs=ctx.get("attachment_objs") or (), clone_index=ctx.get("template_index", 0))
    if spec.get("resourceClaims") and ctx.get("dra_enabled", True):
        # DynamicResources' PreFilter runs after the volume plugins' (default_plugins.go:45-47); the fake cluster holds no ResourceClaim
        if ctx.get("dra_partial"):
            raise NotImplementedError("the scheduler configuration disables only the filter point of DynamicResources: a pod with resourceClaims is not modelled under it")
        if pod.prefilter_reject is None:
            pod.prefilter_reject = V.dra_prefilter(sim_pod, ctx.get("template_index", 0))

    # topology spread constraints (common.go:86-127); NodeAffinityPolicy defaults to Honor, NodeTaintsPolicy to Ignore
    included = np.array([node_matches_required(i) for i in range(N)], np.uint8) if affinity_active else None
    constraints = list(spec.get("topologySpreadConstraints") or [])
    if not constraints and ctx.get("default_spreading") is not None:
        # System default spreading (a Service / the controller selects the template): two more ScheduleAnyway constraints of the pod,
        # scored with requireAllTopologies = false (scoring.go:61-115,140: a node without a key is not ignored, the missing key counts
        # as the value "" when the domains are sized and scores nothing) -- PodSpec.soft_relaxed; the binding derives the engine's form
        # (model.relax_soft).  Left out, and the caller told (pod.default_spreading_unmodelled), only when a node lacks the HOSTNAME
        # label (the per-node constraint has no column to read then) or several templates run.
        defaults = system_default_constraints(sim_pod, *ctx["default_spreading"])
        if defaults:
            # (several templates: the engine keeps each template's spread state apart, a shared Service selector would couple them)
            if ctx["n_templates"] == 1 and all(HOSTNAME in (n["metadata"].get("labels") or {}) for n in nodes):
                constraints = defaults
                pod.soft_relaxed = True
            else:
                ctx["default_spreading_unmodelled"] = True
    for c in constraints:
        sel = c.get("labelSelector")
        # matchLabelKeys (common.go:95-105): the incoming pod's own values of these keys are ANDed into the selector
        merged = [{"key": k, "operator": "In", "values": [sim_labels[k]]} for k in c.get("matchLabelKeys") or [] if k in sim_labels]
        if sel is not None and merged:
            sel = {"matchLabels": dict(sel.get("matchLabels") or {}), "matchExpressions": list(sel.get("matchExpressions") or []) + merged}
        col = it.col(c["topologyKey"])

        def matches(p):  # countPodsMatchSelector (common.go:144-159)
            return (not selector_empty(sel) and (p["metadata"].get("namespace") or "default") == sim_ns
                    and not p["metadata"].get("deletionTimestamp") and label_selector_matches(sel, p["metadata"].get("labels") or {}))
        existing = np.zeros(N, np.int32)
        is_hard = (c.get("whenUnsatisfiable") or "DoNotSchedule") == "DoNotSchedule"
        for p in live:
            if matches(p):
                existing[index[p["spec"]["nodeName"]]] += 1
                if is_hard and id(p) in victim_ids:
                    interacts[index[p["spec"]["nodeName"]]] = 1
        # matchNodeInclusionPolicies (common.go:107-122): required node affinity / selector (default Honor) and the
        # NoSchedule / NoExecute taints the pod does not tolerate (default Ignore)
        honor_aff = (c.get("nodeAffinityPolicy") or "Honor") == "Honor"
        honor_taints = (c.get("nodeTaintsPolicy") or "Ignore") == "Honor"
        inc = None
        if honor_aff and included is not None:
            inc = included.copy()
        if honor_taints:
            tol_ok = np.array(ok, np.uint8)[ts_id]
            inc = tol_ok if inc is None else (inc & tol_ok)
        pod.spread.append(M.SpreadConstraint(
            col=col, max_skew=int32_field(c.get("maxSkew")), min_domains=int32_field(c.get("minDomains") or None, 1),
            hard=(c.get("whenUnsatisfiable") or "DoNotSchedule") == "DoNotSchedule",
            self_match=not selector_empty(sel) and label_selector_matches(sel, sim_labels),
            is_hostname=c["topologyKey"] == HOSTNAME, n_domains=len(it.values[col]),
            node_match_count=existing if existing.any() else None, node_included=inc))

    # inter-pod affinity (filtering.go:204-432, scoring.go:81-125)
    pa = (spec.get("affinity") or {}).get("podAffinity") or {}
    paa = (spec.get("affinity") or {}).get("podAntiAffinity") or {}
    r_aff = pa.get("requiredDuringSchedulingIgnoredDuringExecution") or []
    r_anti = paa.get("requiredDuringSchedulingIgnoredDuringExecution") or []
    p_aff = pa.get("preferredDuringSchedulingIgnoredDuringExecution") or []
    p_anti = paa.get("preferredDuringSchedulingIgnoredDuringExecution") or []
    others_have_terms = any(((p.get("spec") or {}).get("affinity") or {}).get(k) for p in live for k in ("podAffinity", "podAntiAffinity"))
    if r_aff or r_anti or p_aff or p_anti or others_have_terms:
        keys: List[str] = []

        def kidx(k):
            if k not in keys:
                keys.append(k)
            return keys.index(k)

        sim_as_pod = {"metadata": {"namespace": sim_ns, "labels": sim_labels}}
        ipa = M.InterPodAffinity(key_cols=[], key_ndom=[])
        ipa.aff_keys = [kidx(t["topologyKey"]) for t in r_aff]
        ipa.self_aff = bool(r_aff) and all(tm(t, sim_ns, sim_as_pod) for t in r_aff)
        ipa.anti_keys = [kidx(t["topologyKey"]) for t in r_anti]
        ipa.anti_self = [tm(t, sim_ns, sim_as_pod) for t in r_anti]
        aff_existing = np.zeros(N, np.int32)
        anti_existing = [np.zeros(N, np.int32) for _ in r_anti]
        exist_anti: Dict[int, np.ndarray] = {}
        score_existing: Dict[int, np.ndarray] = {}
        entries = 0

        def add_score(k, i, w):
            nonlocal entries
            col = it.col(keys[k])
            if it.arrays[col][i]:
                score_existing.setdefault(k, np.zeros(N, np.int64))[i] += w
                entries += 1

        for p in live:
            i = index[p["spec"]["nodeName"]]
            p_ns = p["metadata"].get("namespace") or "default"
            vic = id(p) in victim_ids
            if r_aff and all(tm(t, sim_ns, p) for t in r_aff):
                aff_existing[i] += 1
                interacts[i] |= vic
            for t_i, t in enumerate(r_anti):
                if tm(t, sim_ns, p):
                    anti_existing[t_i][i] += 1
                    interacts[i] |= vic
            e_aff = ((p["spec"].get("affinity") or {}).get("podAffinity") or {})
            e_anti = ((p["spec"].get("affinity") or {}).get("podAntiAffinity") or {})
            for t in e_anti.get("requiredDuringSchedulingIgnoredDuringExecution") or []:
                if tm(t, p_ns, sim_as_pod):
                    exist_anti.setdefault(kidx(t["topologyKey"]), np.zeros(N, np.int32))[i] += 1
                    interacts[i] |= vic
            # scoring.go:81-125 processExistingPod
            for wt in p_aff:
                if tm(wt["podAffinityTerm"], sim_ns, p):
                    add_score(kidx(wt["podAffinityTerm"]["topologyKey"]), i, int32_field(wt.get("weight")))
            for wt in p_anti:
                if tm(wt["podAffinityTerm"], sim_ns, p):
                    add_score(kidx(wt["podAffinityTerm"]["topologyKey"]), i, -int32_field(wt.get("weight")))
            if hard_pod_affinity_weight > 0:
                for t in e_aff.get("requiredDuringSchedulingIgnoredDuringExecution") or []:
                    if tm(t, p_ns, sim_as_pod):
                        add_score(kidx(t["topologyKey"]), i, hard_pod_affinity_weight)
            for wt in e_aff.get("preferredDuringSchedulingIgnoredDuringExecution") or []:
                if tm(wt["podAffinityTerm"], p_ns, sim_as_pod):
                    add_score(kidx(wt["podAffinityTerm"]["topologyKey"]), i, int32_field(wt.get("weight")))
            for wt in e_anti.get("preferredDuringSchedulingIgnoredDuringExecution") or []:
                if tm(wt["podAffinityTerm"], p_ns, sim_as_pod):
                    add_score(kidx(wt["podAffinityTerm"]["topologyKey"]), i, -int32_field(wt.get("weight")))
        # what ONE clone adds (it is an existing pod of the next cycle, with the incoming pod's own terms)
        self_score: Dict[int, int] = {}   # (a term of the incoming pod may name a topology key no existing pod touched:
        self_hits: Dict[int, int] = {}    #  kidx can still grow here)
        for wt, sign in [(w, 1) for w in p_aff] + [(w, -1) for w in p_anti]:
            if tm(wt["podAffinityTerm"], sim_ns, sim_as_pod):  # both directions: incoming's term vs the
                k = kidx(wt["podAffinityTerm"]["topologyKey"])              # clone, and the clone's term vs the incoming pod
                self_score[k] = self_score.get(k, 0) + 2 * sign * int32_field(wt.get("weight"))
                self_hits[k] = self_hits.get(k, 0) + 2
        if hard_pod_affinity_weight > 0:
            for t in r_aff:
                if tm(t, sim_ns, sim_as_pod):
                    k = kidx(t["topologyKey"])
                    self_score[k] = self_score.get(k, 0) + hard_pod_affinity_weight
                    self_hits[k] = self_hits.get(k, 0) + 1
        score_self = [self_score.get(k, 0) for k in range(len(keys))]
        self_entries = [self_hits.get(k, 0) for k in range(len(keys))]
        if len(keys) > M.MAX_IPA_KEYS:
            raise NotImplementedError("more than %d distinct inter-pod affinity topology keys" % M.MAX_IPA_KEYS)
        ipa.key_cols = [it.col(k) for k in keys]
        ipa.key_ndom = [len(it.values[c]) for c in ipa.key_cols]
        ipa.aff_existing = aff_existing if aff_existing.any() else None
        ipa.anti_existing = [a if a.any() else None for a in anti_existing]
        ipa.exist_anti = [exist_anti.get(k) for k in range(len(keys))]
        ipa.score_existing = [score_existing.get(k) for k in range(len(keys))]
        ipa.score_self, ipa.self_entries, ipa.entries_existing = score_self, self_entries, entries
        pod.ipa = ipa
    pre.victim_interacts = interacts if interacts.any() else None
    return pod, reasons
