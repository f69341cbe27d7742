"""genpod: the pod a namespace's limits describe (cmd/genpod, pkg/client/nspod.go:34-126), from snapshot objects.

Python test mirror of cluster-capacity_amd/host/genpod.hpp.  The reference asks the API server for the Namespace and its
LimitRanges; here they come from the same `--snapshot` files as the nodes (kubectl get ns,limitrange -A -o yaml).
  * resources: for memory, cpu and nvdia.com/gpu (sic, nspod.go:31) the MINIMUM over all LimitRange items of type Pod of
    `max[resource]` (:72-88, Quantity.Cmp); if any of them is non-zero, the stub container gets them as limits AND requests
    (:90-117);
  * nodeSelector: the namespace's `openshift.io/node-selector` annotation, "k1=v1,k2=v2" (:119-126,
    apimachinery labels.ConvertSelectorToLabelsMap labels.go:159-183)."""
from __future__ import annotations

import math
from fractions import Fraction
from typing import List

from . import ingest

RESOURCES = ("memory", "cpu", "nvdia.com/gpu")  # nspod.go:66-70 (the typo is the reference's)


class GenpodError(ValueError):
    pass


def selector_to_labels(selector: str) -> dict:
    out = {}
    if not selector:
        return out
    for label in selector.split(","):
        kv = label.split("=")
        if len(kv) != 2:
            raise GenpodError(f"invalid selector: {kv}")
        out[kv[0].strip()] = kv[1].strip()
    return out


def namespace_pod(namespace: str, namespace_objs: List[dict], limit_range_objs: List[dict]) -> dict:
    ns = next((n for n in namespace_objs if n["metadata"]["name"] == namespace), None)
    if ns is None:
        raise GenpodError(f"Namespace {namespace} not found")
    pod = {"apiVersion": "v1", "kind": "Pod",
           "metadata": {"name": "cluster-capacity-stub-container", "namespace": namespace},
           "spec": {"containers": [{"name": "cluster-capacity-stub-container", "image": "gcr.io/google_containers/pause:2.0",
                                    "imagePullPolicy": "Always"}],
                    "restartPolicy": "OnFailure", "dnsPolicy": "Default"}}
    best = {}
    for lr in limit_range_objs:
        if (lr["metadata"].get("namespace") or "default") != namespace:
            continue
        for item in (lr.get("spec") or {}).get("limits") or []:
            if item.get("type") != "Pod":
                continue
            for r in RESOURCES:
                if r not in (item.get("max") or {}):
                    continue
                amount = str(item["max"][r])
                if r not in best or ingest.parse_quantity(best[r]) > ingest.parse_quantity(amount):
                    best[r] = amount
    if any(ingest.parse_quantity(q) != 0 for q in best.values()):
        # (the serializer prints the Quantity, not the text it was read from: "0.5" leaves as "500m", "1024Mi" as "1Gi")
        canon = {r: ingest.quantity_canonical(Fraction(math.ceil(ingest.parse_quantity(q) * 10**9), 10**9), ingest.quantity_format(q)) for r, q in best.items()}
        pod["spec"]["containers"][0]["resources"] = {"limits": dict(canon), "requests": dict(canon)}
    ann = ns["metadata"].get("annotations") or {}
    if "openshift.io/node-selector" in ann:
        try:
            pod["spec"]["nodeSelector"] = selector_to_labels(ann["openshift.io/node-selector"])
        except GenpodError as e:
            raise GenpodError(f"Unable to parse openshift.io/node-selector in {ann['openshift.io/node-selector']} namespace: {e}")
    return pod
