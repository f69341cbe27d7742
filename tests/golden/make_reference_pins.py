"""Regenerates tests/golden/reference_pins.json: the literal strings, status codes and constants of the path, READ OUT OF THE
REFERENCE'S SOURCES (/root/reference, Go -- it cannot be executed in the build image, but it can be read).

    python tests/golden/make_reference_pins.py

Everything the hosts print and every constant the oracle / the kernels hard-code is a transcription; this pins the transcriptions
to the files they came from (tests/test_reference_pins.py compares the hosts, the oracle and the engine's constants with the
fixture, and -- where /root/reference is present -- the fixture with the sources again).  Each entry records file, line and value."""
import json
import os
import re

REF = os.environ.get("CC_REFERENCE", "/root/reference")
S = "vendor/k8s.io/kubernetes/pkg/scheduler"
P = S + "/framework/plugins"

# name -> (file, regex with one group)
STRINGS = {
    "reason.unschedulable": (P + "/nodeunschedulable/node_unschedulable.go", r'ErrReasonUnschedulable = "([^"]+)"'),
    "reason.nodename": (P + "/nodename/node_name.go", r'ErrReason = "([^"]+)"'),
    "reason.nodeaffinity": (P + "/nodeaffinity/node_affinity.go", r'ErrReasonPod = "([^"]+)"'),
    "reason.nodeports": (P + "/nodeports/node_ports.go", r'ErrReason = "([^"]+)"'),
    "reason.too_many_pods": (P + "/noderesources/fit.go", r'Reason:\s+"(Too many pods)"'),
    "reason.insufficient_cpu": (P + "/noderesources/fit.go", r'Reason:\s+"(Insufficient cpu)"'),
    "reason.insufficient_memory": (P + "/noderesources/fit.go", r'Reason:\s+"(Insufficient memory)"'),
    "reason.insufficient_ephemeral": (P + "/noderesources/fit.go", r'Reason:\s+"(Insufficient ephemeral-storage)"'),
    "reason.insufficient_scalar_format": (P + "/noderesources/fit.go", r'Reason:\s+fmt\.Sprintf\("(Insufficient %v)", rName\)'),
    "reason.pts_skew": (P + "/podtopologyspread/plugin.go", r'ErrReasonConstraintsNotMatch = "([^"]+)"'),
    "reason.pts_missing_label_suffix": (P + "/podtopologyspread/plugin.go", r'ErrReasonNodeLabelNotMatch = ErrReasonConstraintsNotMatch \+ "([^"]+)"'),
    "reason.ipa_existing_anti": (P + "/interpodaffinity/filtering.go", r'ErrReasonExistingAntiAffinityRulesNotMatch = "([^"]+)"'),
    "reason.ipa_affinity": (P + "/interpodaffinity/filtering.go", r'ErrReasonAffinityRulesNotMatch = "([^"]+)"'),
    "reason.ipa_anti": (P + "/interpodaffinity/filtering.go", r'ErrReasonAntiAffinityRulesNotMatch = "([^"]+)"'),
    "reason.taint_format": (P + "/tainttoleration/taint_toleration.go", r'fmt\.Sprintf\("(node\(s\) had untolerated taint \{%s: %s\})"'),
    "fit_error.prefix_format": (S + "/framework/types.go", r'NoNodeAvailableMsg = "([^"]+)"'),
    "preemption.no_victims": (P + "/defaultpreemption/default_preemption.go", r'"(No preemption victims found for incoming pod)"'),
    "preemption.not_helpful": (S + "/framework/preemption/preemption.go", r'"(Preemption is not helpful for scheduling)"'),
    "preemption.never": (P + "/defaultpreemption/default_preemption.go", r'"(not eligible due to preemptionPolicy=Never\.)"'),
    "preemption.prefix": (P + "/defaultpreemption/default_preemption.go", r'"(preemption: )"\+msg'),
    "resource.prefix_native": ("vendor/k8s.io/api/core/v1/types.go", r'ResourceDefaultNamespacePrefix = "([^"]+)"'),
    "resource.prefix_hugepages": ("vendor/k8s.io/api/core/v1/types.go", r'ResourceHugePagesPrefix = "([^"]+)"'),
    "resource.prefix_attachable": ("vendor/k8s.io/api/core/v1/types.go", r'ResourceAttachableVolumesPrefix = "([^"]+)"'),
    "resource.prefix_requests": ("vendor/k8s.io/api/core/v1/types.go", r'DefaultResourceRequestsPrefix = "([^"]+)"'),
    "qname.char": ("vendor/k8s.io/apimachinery/pkg/util/validation/validation.go", r'const qnameCharFmt string = "([^"]+)"'),
    "qname.ext_char": ("vendor/k8s.io/apimachinery/pkg/util/validation/validation.go", r'const qnameExtCharFmt string = "([^"]+)"'),
    "qname.dns1123_label": ("vendor/k8s.io/apimachinery/pkg/util/validation/validation.go", r'const dns1123LabelFmt string = "([^"]+)"'),
    "report.gpu_resource_name": ("pkg/framework/report.go", r'ResourceNvidiaGPU v1\.ResourceName = "([^"]+)"'),
    "toleration.op_exists": ("vendor/k8s.io/api/core/v1/types.go", r'TolerationOpExists TolerationOperator = "([^"]+)"'),
    "toleration.op_equal": ("vendor/k8s.io/api/core/v1/types.go", r'TolerationOpEqual\s+TolerationOperator = "([^"]+)"'),
    "label.zone": ("vendor/k8s.io/api/core/v1/well_known_labels.go", r'LabelTopologyZone\s+= "([^"]+)"'),
    "label.region": ("vendor/k8s.io/api/core/v1/well_known_labels.go", r'LabelTopologyRegion = "([^"]+)"'),
    "label.zone_beta": ("vendor/k8s.io/api/core/v1/well_known_labels.go", r'LabelFailureDomainBetaZone\s+= "([^"]+)"'),
    "label.region_beta": ("vendor/k8s.io/api/core/v1/well_known_labels.go", r'LabelFailureDomainBetaRegion = "([^"]+)"'),
    "stop.limit_format": ("pkg/framework/simulator.go", r'fmt\.Sprintf\("(LimitReached: Maximum number of pods simulated: %v)"'),
    "report.headline_format": ("pkg/framework/report.go", r'fmt\.Printf\("(The cluster can schedule %v instance\(s\) of the pod %v\.)\\n"'),
    "report.termination_format": ("pkg/framework/report.go", r'fmt\.Printf\("\\n(Termination reason: %v: %v)\\n"'),
    "report.distribution_header": ("pkg/framework/report.go", r'fmt\.Printf\("\\n(Pod distribution among nodes:)\\n"\)'),
    "report.node_line_format": ("pkg/framework/report.go", r'fmt\.Printf\("\\t(- %v: %v instance\(s\))\\n"'),
    "report.requirements_format": ("pkg/framework/report.go", r'fmt\.Printf\("(%v pod requirements:)\\n"'),
}
# status code of a filter failure: name -> (file, regex capturing the code next to the reason constant)
CODES = {
    "code.unschedulable": (P + "/nodeunschedulable/node_unschedulable.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonUnschedulable\)'),
    "code.nodename": (P + "/nodename/node_name.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReason\)'),
    "code.nodeaffinity": (P + "/nodeaffinity/node_affinity.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonPod\)'),
    "code.nodeports": (P + "/nodeports/node_ports.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReason\)'),
    "code.pts_missing_label": (P + "/podtopologyspread/filtering.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonNodeLabelNotMatch\)'),
    "code.pts_skew": (P + "/podtopologyspread/filtering.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonConstraintsNotMatch\)'),
    "code.ipa_affinity": (P + "/interpodaffinity/filtering.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonAffinityRulesNotMatch\)'),
    "code.ipa_anti": (P + "/interpodaffinity/filtering.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonAntiAffinityRulesNotMatch\)'),
    "code.ipa_existing_anti": (P + "/interpodaffinity/filtering.go", r'fwk\.NewStatus\(fwk\.(\w+), ErrReasonExistingAntiAffinityRulesNotMatch\)'),
    "code.taint": (P + "/tainttoleration/taint_toleration.go", r'fwk\.NewStatus\(fwk\.(\w+), errReason\)'),
    "code.fit_default": (P + "/noderesources/fit.go", r'statusCode := fwk\.(\w+)\n'),
    "code.fit_beyond_allocatable": (P + "/noderesources/fit.go", r'if insufficientResources\[i\]\.Unresolvable \{\n\t+statusCode = fwk\.(\w+)'),
    "code.preemption_no_victims": (P + "/defaultpreemption/default_preemption.go", r'fwk\.NewStatus\(fwk\.(\w+), "No preemption victims found for incoming pod"\)'),
}
# integer constants: name -> (file, regex capturing a Go integer expression)
NUMBERS = {
    "weight.TaintToleration": (S + "/apis/config/v1/default_plugins.go", r'names\.TaintToleration, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "weight.NodeAffinity": (S + "/apis/config/v1/default_plugins.go", r'names\.NodeAffinity, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "weight.NodeResourcesFit": (S + "/apis/config/v1/default_plugins.go", r'names\.NodeResourcesFit, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "weight.PodTopologySpread": (S + "/apis/config/v1/default_plugins.go", r'names\.PodTopologySpread, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "weight.InterPodAffinity": (S + "/apis/config/v1/default_plugins.go", r'names\.InterPodAffinity, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "weight.NodeResourcesBalancedAllocation": (S + "/apis/config/v1/default_plugins.go", r'names\.NodeResourcesBalancedAllocation, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "weight.ImageLocality": (S + "/apis/config/v1/default_plugins.go", r'names\.ImageLocality, Weight: ptr\.To\[int32\]\((\d+)\)'),
    "default.milli_cpu_request": (S + "/util/pod_resources.go", r'DefaultMilliCPURequest int64 = ([\d \*]+)'),
    "default.memory_request": (S + "/util/pod_resources.go", r'DefaultMemoryRequest int64 = ([\d \*]+)'),
    "search.min_feasible_nodes": (S + "/schedule_one.go", r'\n\tminFeasibleNodesToFind = (\d+)'),
    "search.min_feasible_percentage": (S + "/schedule_one.go", r'\n\tminFeasibleNodesPercentageToFind = (\d+)'),
    "default.percentage_of_nodes_to_score": (S + "/apis/config/types.go", r'\n\tDefaultPercentageOfNodesToScore = (\d+)'),
    "default.hard_pod_affinity_weight": (S + "/apis/config/v1/defaults.go", r'obj\.HardPodAffinityWeight = ptr\.To\[int32\]\((\d+)\)'),
    "qname.max_length": ("vendor/k8s.io/apimachinery/pkg/util/validation/validation.go", r'const qualifiedNameMaxLength int = (\d+)'),
    "qname.dns1123_subdomain_max_length": ("vendor/k8s.io/apimachinery/pkg/util/validation/validation.go", r'const DNS1123SubdomainMaxLength int = (\d+)'),
    "score.max_node_score": (S + "/framework/interface.go", r'MaxNodeScore int64 = (\d+)'),
    "image.mb": (P + "/imagelocality/image_locality.go", r'\n\tmb\s+int64 = ([\d \*]+)'),
    "image.min_threshold_mb": (P + "/imagelocality/image_locality.go", r'minThreshold\s+int64 = (\d+) \* mb'),
    "image.max_container_threshold_mb": (P + "/imagelocality/image_locality.go", r'maxContainerThreshold int64 = (\d+) \* mb'),
}


def _find(rel, pattern):
    text = open(os.path.join(REF, rel)).read()
    m = re.search(pattern, text)
    if not m:
        raise SystemExit(f"{rel}: pattern not found: {pattern}")
    return m.group(1), text.count("\n", 0, m.start(1)) + 1


def plugin_order():
    """The default MultiPoint plugin list, in order (apis/config/v1/default_plugins.go getDefaultPlugins): Filter plugins run in this
    order and the first failing one decides a node's reasons and status code."""
    rel = S + "/apis/config/v1/default_plugins.go"
    text = open(os.path.join(REF, rel)).read()
    body = text[text.index("func getDefaultPlugins()"):text.index("applyFeatureGates(plugins)")]
    return {"value": re.findall(r"\{Name: names\.(\w+)", body), "file": rel, "line": text.count("\n", 0, text.index("func getDefaultPlugins()")) + 1}


def collect():
    out = {"plugins.multipoint_order": plugin_order()}
    for table, conv in ((STRINGS, str), (CODES, str), (NUMBERS, lambda e: int(eval(e, {"__builtins__": {}})))):
        for name, (rel, pattern) in table.items():
            value, line = _find(rel, pattern)
            out[name] = {"value": conv(value.strip() if table is NUMBERS else value), "file": rel, "line": line}
    return out


if __name__ == "__main__":
    pins = collect()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_pins.json")
    json.dump(pins, open(path, "w"), indent=1, sort_keys=True)
    print(f"{len(pins)} pins -> {path}")
