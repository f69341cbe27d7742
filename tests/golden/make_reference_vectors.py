"""Regenerates tests/golden/reference_vectors.json: input/output vectors of the reference's OWN arithmetic, obtained by executing a
mechanical transliteration of its Go source text.

    python tests/golden/make_reference_vectors.py

The reference is Go and there is no Go toolchain in the build image, so it cannot be run as a whole.  Its scoring arithmetic,
however, lives in a handful of small pure functions written in a subset of Go that maps line by line onto Python: int64 / float64
arithmetic, if / else, for-range loops, append, math.Abs / math.Sqrt.  This script reads those functions out of /root/reference
(file + line recorded), rewrites them token by token (rules below: no understanding of what the code does is involved, and none
is injected), executes the result on adversarial + random inputs, and writes inputs and outputs to the fixture.
tests/test_reference_vectors.py then holds the oracle's unit functions (oracle/ccref.c) against the fixture.  Where Python and Go
semantics differ the rules say so: integer division truncates in Go (godiv), int64(float) truncates (goint), and the functions
that divide integers are listed (INT_DIV) -- `/` between float64 operands is Python's `/`.  math.Log is NOT covered: Go's amd64
implementation is its own (the oracle restates it; no other implementation of that polynomial exists here to execute).

Functions: leastRequestedScore + the closure of leastResourceScorer (noderesources/least_allocated.go), balancedResourceScorer
(noderesources/balanced_allocation.go), DefaultNormalizeScore (helper/normalize_score.go), numFeasibleNodesToFind (schedule_one.go),
calculatePriority + scaledImageScore (imagelocality/image_locality.go), scoreForCount (podtopologyspread/scoring.go); and the string-level
helpers of FUNCS' second block (toleration / taint verdicts, zone key, image names, label requirements), fitsRequest (noderesources/fit.go) and
InterPodAffinity's Filter with its three satisfy* functions (interpodaffinity/filtering.go), PodTopologySpread's Filter with minMatchNum
(podtopologyspread/filtering.go),
run on Python objects that carry the Go method surface (GoNodeInfo, GoResource, GoPodRequest below)."""
import json
import math
import os
import random
import re
import types

REF = os.environ.get("CC_REFERENCE", "/root/reference")
S = "vendor/k8s.io/kubernetes/pkg/scheduler"
HERE = os.path.dirname(os.path.abspath(__file__))
PINS = {k: v["value"] for k, v in json.load(open(os.path.join(HERE, "reference_pins.json"))).items()}

# (name, file, the line that starts the body to cut, parameter names, does `/` divide integers?)
FUNCS = [
    ("leastRequestedScore", S + "/framework/plugins/noderesources/least_allocated.go", "func leastRequestedScore(requested, capacity int64) int64 {", ["requested", "capacity"], True),
    ("leastResourceScorer_closure", S + "/framework/plugins/noderesources/least_allocated.go", "\treturn func(requested, allocable []int64) int64 {", ["requested", "allocable", "resources"], True),
    ("balancedResourceScorer", S + "/framework/plugins/noderesources/balanced_allocation.go", "func balancedResourceScorer(requested, allocable []int64) int64 {", ["requested", "allocable"], False),
    ("DefaultNormalizeScore", S + "/framework/plugins/helper/normalize_score.go",
     "func DefaultNormalizeScore(maxPriority int64, reverse bool, scores framework.NodeScoreList) *fwk.Status {", ["maxPriority", "reverse", "scores"], True),
    ("numFeasibleNodesToFind", S + "/schedule_one.go",
     "func (sched *Scheduler) numFeasibleNodesToFind(percentageOfNodesToScore *int32, numAllNodes int32) (numNodes int32) {", ["sched", "percentageOfNodesToScore", "numAllNodes"], True),
    ("calculatePriority", S + "/framework/plugins/imagelocality/image_locality.go", "func calculatePriority(sumScores int64, numContainers int) int64 {", ["sumScores", "numContainers"], True),
    ("scaledImageScore", S + "/framework/plugins/imagelocality/image_locality.go", "func scaledImageScore(imageState *fwk.ImageStateSummary, totalNumNodes int) int64 {",
     ["imageState", "totalNumNodes"], False),
    ("scoreForCount", S + "/framework/plugins/podtopologyspread/scoring.go", "func scoreForCount(cnt int64, maxSkew int32, tpWeight float64) float64 {", ["cnt", "maxSkew", "tpWeight"], False),
    # the two plugin NormalizeScore methods: the lines that fetch the cycle state (getPreScoreState ... return nil) are dropped (DROP),
    # `s.IgnoredNodes.Has(score.Name)` reads the `ignored` argument, `len(s.topologyScore) == 0` is the caller's business
    ("ptsNormalizeScore", S + "/framework/plugins/podtopologyspread/scoring.go",
     "func (pl *PodTopologySpread) NormalizeScore(ctx context.Context, cycleState fwk.CycleState, pod *v1.Pod, scores framework.NodeScoreList) *fwk.Status {",
     ["scores", "ignored"], True),
    ("ipaNormalizeScore", S + "/framework/plugins/interpodaffinity/scoring.go",
     "func (pl *InterPodAffinity) NormalizeScore(ctx context.Context, cycleState fwk.CycleState, pod *v1.Pod, scores framework.NodeScoreList) *fwk.Status {",
     ["scores"], False),
]
# string-level helpers the ingests mirror: toleration matching, the zone key of the node tree, image name normalisation
FUNCS += [
    ("ToleratesTaint", "vendor/k8s.io/api/core/v1/toleration.go", "func (t *Toleration) ToleratesTaint(taint *Taint) bool {", ["t", "taint"], False),
    ("GetZoneKey", "vendor/k8s.io/component-helpers/node/topology/helpers.go", "func GetZoneKey(node *v1.Node) string {", ["node"], False),
    ("normalizedImageName", S + "/framework/plugins/imagelocality/image_locality.go", "func normalizedImageName(name string) string {", ["name"], False),
    # label / node-selector requirements (NodeAffinity, nodeSelector, label selectors): In NotIn Exists DoesNotExist Gt Lt.  strconv.ParseInt is
    # Go's standard library (not in the tree): parse_int below stands in for it; the klog lines are dropped
    ("requirementHasValue", "vendor/k8s.io/apimachinery/pkg/labels/selector.go", "func (r *Requirement) hasValue(value string) bool {", ["r", "value"], False),
    ("requirementMatches", "vendor/k8s.io/apimachinery/pkg/labels/selector.go", "func (r *Requirement) Matches(ls Labels) bool {", ["r", "ls"], False),
    # the TaintToleration plugin's verdicts: the Filter's first untolerated NoSchedule / NoExecute taint, the Score's count of untolerated
    # PreferNoSchedule taints over the tolerations PreScore keeps
    ("TolerationsTolerateTaint", "vendor/k8s.io/component-helpers/scheduling/corev1/helpers.go",
     "func TolerationsTolerateTaint(tolerations []v1.Toleration, taint *v1.Taint) bool {", ["tolerations", "taint"], False),
    ("getFilteredTaints", "vendor/k8s.io/component-helpers/scheduling/corev1/helpers.go",
     "func getFilteredTaints(taints []v1.Taint, inclusionFilter taintsFilterFunc) []v1.Taint {", ["taints", "inclusionFilter"], False),
    ("FindMatchingUntoleratedTaint", "vendor/k8s.io/component-helpers/scheduling/corev1/helpers.go",
     "func FindMatchingUntoleratedTaint(taints []v1.Taint, tolerations []v1.Toleration, inclusionFilter taintsFilterFunc) (v1.Taint, bool) {",
     ["taints", "tolerations", "inclusionFilter"], False),
    ("DoNotScheduleTaintsFilter_closure", S + "/framework/plugins/helper/taint.go", "\treturn func(t *v1.Taint) bool {", ["t"], False),
    ("getAllTolerationPreferNoSchedule", S + "/framework/plugins/tainttoleration/taint_toleration.go",
     "func getAllTolerationPreferNoSchedule(tolerations []v1.Toleration) (tolerationList []v1.Toleration) {", ["tolerations"], False),
    # the NodeResourcesFit filter itself.  nodeInfo / podRequest are Python objects with the Go method surface (GetAllocatable().GetMilliCPU() ...);
    # Go ranges over the scalar map in random order, here sorted (the reasons of scalars are compared as a set)
    ("fitsRequest", S + "/framework/plugins/noderesources/fit.go",
     "func fitsRequest(podRequest *preFilterState, nodeInfo fwk.NodeInfo, ignoredExtendedResources, ignoredResourceGroups sets.Set[string], opts ResourceRequestsOptions) []InsufficientResource {",
     ["podRequest", "nodeInfo", "ignoredExtendedResources", "ignoredResourceGroups", "opts"], False),
    # PodTopologySpread's Filter over the per-constraint domain counts PreFilter built (TpValueToMatchNum and the critical paths are put together by the
    # harness; the local `minMatchNum` shadows the method's name in Python, hence the prefixed function name)
    ("preFilterState_minMatchNum", S + "/framework/plugins/podtopologyspread/filtering.go", "func (s *preFilterState) minMatchNum(constraintID int, minDomains int32) (int, error) {", ["s", "constraintID", "minDomains"], False),
    ("ptsFilter", S + "/framework/plugins/podtopologyspread/filtering.go", "func (pl *PodTopologySpread) Filter(ctx context.Context, cycleState fwk.CycleState, pod *v1.Pod, nodeInfo fwk.NodeInfo) *fwk.Status {", ["s", "node", "pod"], False),
    # InterPodAffinity's Filter over the three count maps PreFilter built (the maps themselves are put together by the harness)
    ("satisfyExistingPodsAntiAffinity", S + "/framework/plugins/interpodaffinity/filtering.go", "func satisfyExistingPodsAntiAffinity(state *preFilterState, nodeInfo fwk.NodeInfo) bool {", ["state", "nodeInfo"], False),
    ("satisfyPodAntiAffinity", S + "/framework/plugins/interpodaffinity/filtering.go", "func satisfyPodAntiAffinity(state *preFilterState, nodeInfo fwk.NodeInfo) bool {", ["state", "nodeInfo"], False),
    ("satisfyPodAffinity", S + "/framework/plugins/interpodaffinity/filtering.go", "func satisfyPodAffinity(state *preFilterState, nodeInfo fwk.NodeInfo) bool {", ["state", "nodeInfo"], False),
    ("ipaFilter", S + "/framework/plugins/interpodaffinity/filtering.go", "func (pl *InterPodAffinity) Filter(ctx context.Context, cycleState fwk.CycleState, pod *v1.Pod, nodeInfo fwk.NodeInfo) *fwk.Status {", ["state", "nodeInfo"], False),
    ("countIntolerableTaintsPreferNoSchedule", S + "/framework/plugins/tainttoleration/taint_toleration.go",
     "func countIntolerableTaintsPreferNoSchedule(taints []v1.Taint, tolerations []v1.Toleration) (intolerableTaints int) {", ["taints", "tolerations"], False),
]
# Round 3 -- the LOOP-LEVEL pieces the oracle restates by hand: the counting loop of PodTopologySpread's PreFilter, the
# updates of InterPodAffinity's topology-pair maps, the weight-and-sum block of RunScorePlugins, selectHost, topologyNormalizingWeight.
# Entries may carry a 6th element: {"nth": which occurrence of the start line, "block": the cut text is a statement block (a loop), not a function
# body, "header": lines of the function header}.  A closure handed to the parallelizer (`processNode := func(n int) {`, `Until(ctx, len(nodes),
# func(index int) {`) is cut as a function of its index; the harness calls it for 0..n-1 in order (the pieces write disjoint slots).
I = S + "/framework/plugins/interpodaffinity/filtering.go"
IS = S + "/framework/plugins/interpodaffinity/scoring.go"
KT = "vendor/k8s.io/kube-scheduler/framework/types.go"
P = S + "/framework/plugins/podtopologyspread"
FUNCS += [
    ("topologyToMatchedTermCount_update", I, "func (m topologyToMatchedTermCount) update(node *v1.Node, tk string, value int64) {", ["m", "node", "tk", "value"], False),
    ("updateWithAffinityTerms", I, "func (m topologyToMatchedTermCount) updateWithAffinityTerms(", ["m", "terms", "pod", "node", "value"], False, {"header": 2}),
    ("updateWithAntiAffinityTerms", I, "func (m topologyToMatchedTermCount) updateWithAntiAffinityTerms(terms []fwk.AffinityTerm, pod *v1.Pod, nsLabels labels.Set, node *v1.Node, value int64) {",
     ["m", "terms", "pod", "nsLabels", "node", "value"], False),
    ("nodeLabelsMatchSpreadConstraints", P + "/common.go", "func nodeLabelsMatchSpreadConstraints(nodeLabels map[string]string, constraints []topologySpreadConstraint) bool {", ["nodeLabels", "constraints"], False),
    ("countPodsMatchSelector", P + "/common.go", "func countPodsMatchSelector(podInfos []fwk.PodInfo, selector labels.Selector, ns string) int {", ["podInfos", "selector", "ns"], False),
    ("matchNodeInclusionPolicies", P + "/common.go", "func (tsc *topologySpreadConstraint) matchNodeInclusionPolicies(pod *v1.Pod, node *v1.Node, require nodeaffinity.RequiredNodeAffinity) bool {",
     ["tsc", "pod", "node", "require"], False),
    ("criticalPaths_update", P + "/filtering.go", "func (p *criticalPaths) update(tpVal string, num int) {", ["p", "tpVal", "num"], False),
    # calPreFilterState (filtering.go:235-308) in its three pieces: the per-node closure, the merge of the per-node counts, the critical paths
    ("calPreFilterState_processNode", P + "/filtering.go", "\tprocessNode := func(n int) {", ["n", "pl", "pod", "allNodes", "constraints", "requiredNodeAffinity", "tpCountsByNode"], False),
    ("calPreFilterState_merge", P + "/filtering.go", "\tfor _, tpCounts := range tpCountsByNode {", ["s", "tpCountsByNode"], False, {"block": True}),
    ("calPreFilterState_minima", P + "/filtering.go", "\tfor i := 0; i < len(constraints); i++ {", ["s", "constraints"], False, {"block": True, "nth": 1}),
    # RunScorePlugins' second parallel block: plugin weight x normalized score, summed per node (runtime/framework.go:1214-1238)
    ("RunScorePlugins_weigh", S + "/framework/runtime/framework.go", "\tf.Parallelizer().Until(ctx, len(nodes), func(index int) {",
     ["index", "f", "nodes", "plugins", "pluginToNodeScores", "allNodePluginScores", "errCh", "cancel"], False),
    ("nodeScoreHeap_Less", S + "/schedule_one.go", "func (h nodeScoreHeap) Less(i, j int) bool { return h[i].TotalScore > h[j].TotalScore }", ["h", "i", "j"], False, {"oneline": True}),
    ("selectHost", S + "/schedule_one.go", "func selectHost(nodeScoreList []framework.NodePluginScores, count int) (string, []framework.NodePluginScores, error) {", ["nodeScoreList", "count"], False),
    ("topologyNormalizingWeight", P + "/scoring.go", "func topologyNormalizingWeight(size int) float64 {", ["size"], False),
    # PodTopologySpread's PreScore and Score (scoring.go:61-223): initPreScoreState's loop over the FILTERED nodes (ignored nodes, the candidate
    # domains of every constraint and their number), the weights, PreScore's closure over ALL nodes (matching pods counted into the candidate
    # domains), and Score.  *int64 map values are one-element lists here (new(int64) -> [0], atomic.AddInt64(p, n) -> p[0] += n, *p -> p[0])
    ("ptsPreScore_initNodes", P + "/scoring.go", "\tfor _, node := range filteredNodes {", ["s", "filteredNodes", "requireAllTopologies", "topoSize"], False, {"block": True}),
    ("ptsPreScore_weights", P + "/scoring.go", "\tfor i, c := range s.Constraints {", ["s", "filteredNodes", "topoSize"], False, {"block": True}),
    ("ptsPreScore_processAllNode", P + "/scoring.go", "\tprocessAllNode := func(n int) {", ["n", "pl", "pod", "allNodes", "state", "requireAllTopologies", "requiredNodeAffinity"], False),
    # InterPodAffinity's PreScore and Score (interpodaffinity/scoring.go:51-257): the score map's processTerm / processTerms / append, what one
    # existing pod contributes (processExistingPod), PreScore's closure over the nodes, Score.  (`topoScores[atomic.AddInt32(&index, 1)] = x`
    # appends; the merge loop `for i := 0; i <= int(index); i++ { state.topologyScore.append(topoScores[i]) }` is written out by the harness)
    ("scoreMap_processTerm", IS, "func (m scoreMap) processTerm(term *fwk.AffinityTerm, weight int32, pod *v1.Pod, nsLabels labels.Set, node *v1.Node, multiplier int32) {",
     ["m", "term", "weight", "pod", "nsLabels", "node", "multiplier"], False),
    ("scoreMap_processTerms", IS, "func (m scoreMap) processTerms(terms []fwk.WeightedAffinityTerm, pod *v1.Pod, nsLabels labels.Set, node *v1.Node, multiplier int32) {",
     ["m", "terms", "pod", "nsLabels", "node", "multiplier"], False),
    ("scoreMap_append", IS, "func (m scoreMap) append(other scoreMap) {", ["m", "other"], False),
    ("ipa_processExistingPod", IS, "func (pl *InterPodAffinity) processExistingPod(", ["pl", "state", "existingPod", "existingPodNodeInfo", "incomingPod", "topoScore"], False, {"header": 7}),
    ("ipaPreScore_processNode", IS, "\tprocessNode := func(i int) {", ["i", "pl", "allNodes", "hasConstraints", "state", "pod", "topoScores"], False),
    ("ipaScore", IS, "func (pl *InterPodAffinity) Score(ctx context.Context, cycleState fwk.CycleState, pod *v1.Pod, nodeInfo fwk.NodeInfo) (int64, *fwk.Status) {",
     ["s", "node"], False),
    # the node search (schedule_one.go:610-693): the closure the parallelizer runs per visiting position.  The canonical mode of the oracle is ONE
    # worker taking the positions in order and stopping once the context is cancelled -- the harness's loop; feasibleNodesLen is a counter
    # object (atomic.AddInt32(&x, n) -> x.add(n), returning the new value); the two lines that move nextStartNodeIndex (:538-539) are asserted
    # to be in the source and written out by the harness
    ("findNodesThatPassFilters_checkNode", S + "/schedule_one.go", "\tcheckNode := func(i int) {",
     ["i", "sched", "nodes", "numAllNodes", "schedFramework", "ctx", "state", "pod", "errCh", "cancel", "feasibleNodesLen", "numNodesToFind", "feasibleNodes", "result"], False),
    # NodePorts (nodeports/node_ports.go:176-185 over HostPortInfo, kube-scheduler/framework/types.go:427-538): sanitize, NewProtocolPort,
    # CheckConflict, fitsPorts.  *string parameters are one-element lists; the map key *pp is the (protocol, port) pair; HostPortInfo.Add's
    # insertion is the harness's (it calls the transliterated sanitize)
    ("HostPortInfo_sanitize", KT, "func (h HostPortInfo) sanitize(ip, protocol *string) {", ["h", "ip", "protocol"], False),
    ("NewProtocolPort", KT, "func NewProtocolPort(protocol string, port int32) *ProtocolPort {", ["protocol", "port"], False),
    ("HostPortInfo_CheckConflict", KT, "func (h HostPortInfo) CheckConflict(ip, protocol string, port int32) bool {", ["h", "ip", "protocol", "port"], False),
    ("fitsPorts", S + "/framework/plugins/nodeports/node_ports.go", "func fitsPorts(wantPorts []v1.ContainerPort, nodeInfo fwk.NodeInfo) bool {", ["wantPorts", "nodeInfo"], False),
    ("ptsScore", P + "/scoring.go", "func (pl *PodTopologySpread) Score(ctx context.Context, cycleState fwk.CycleState, pod *v1.Pod, nodeInfo fwk.NodeInfo) (int64, *fwk.Status) {",
     ["s", "node", "nodeInfo", "pod"], False),
]
# Round 6 -- the volume plugins' Filters (the object side of cluster-capacity_amd/volumes.py and host/volumes.hpp was compared
# with itself only).  VolumeRestrictions' disk conflicts (volume_restrictions.go:103-160, 266-280), VolumeZone's label match (volume_zone.go:91-100,
# 191-240; the lines that fetch the PreFilter state are dropped: podPVTopologies is an argument), NodeVolumeLimits' counting (csi.go:255-343, 574-586;
# the listers behind pl.* are the harness's), VolumeBinding's bound claims (binder.go:830-865; pvCache / csiNodeLister / CheckNodeAffinity are the harness's).
VP = S + "/framework/plugins"
FUNCS += [
    ("haveOverlap", VP + "/volumerestrictions/volume_restrictions.go", "func haveOverlap(a1, a2 []string) bool {", ["a1", "a2"], False),
    ("isVolumeConflict", VP + "/volumerestrictions/volume_restrictions.go", "func isVolumeConflict(volume *v1.Volume, pod *v1.Pod) bool {", ["volume", "pod"], False),
    ("needsRestrictionsCheck", VP + "/volumerestrictions/volume_restrictions.go", "func needsRestrictionsCheck(v v1.Volume) bool {", ["v"], False),
    ("satisfyVolumeConflicts", VP + "/volumerestrictions/volume_restrictions.go", "func satisfyVolumeConflicts(pod *v1.Pod, nodeInfo fwk.NodeInfo) bool {", ["pod", "nodeInfo"], False),
    ("translateToGALabel", VP + "/volumezone/volume_zone.go", "func translateToGALabel(label string) string {", ["label"], False),
    ("volumeZoneFilter", VP + "/volumezone/volume_zone.go", "func (pl *VolumeZone) Filter(ctx context.Context, cs fwk.CycleState, pod *v1.Pod, nodeInfo fwk.NodeInfo) *fwk.Status {",
     ["podPVTopologies", "pod", "nodeInfo"], False),
    ("getVolumeLimits", VP + "/nodevolumelimits/csi.go", "func getVolumeLimits(csiNode *storagev1.CSINode) map[string]int64 {", ["csiNode"], False),
    ("csiLimitsFilter", VP + "/nodevolumelimits/csi.go", "func (pl *CSILimits) Filter(ctx context.Context, _ fwk.CycleState, pod *v1.Pod, nodeInfo fwk.NodeInfo) *fwk.Status {",
     ["pl", "pod", "nodeInfo"], False),
    ("checkBoundClaims", VP + "/volumebinding/binder.go",
     "func (b *volumeBinder) checkBoundClaims(logger klog.Logger, claims []*v1.PersistentVolumeClaim, node *v1.Node, pod *v1.Pod) (bool, bool, error) {", ["b", "claims", "node", "pod"], False),
]
# per-function textual substitutions applied to a Go statement before the general rules (method calls on receivers the harness models as plain
# Python values; Go's value semantics where Python would alias)
REWRITE = {
    "haveOverlap": [(r"^m := sets\.New\(a1\.\.\.\)$", "m := GoSet(a1)"), (r"^if _, ok := m\[val\]; ok \{$", "if m.Has(val) {")],
    "volumeZoneFilter": [(r"^v, ok = node\.Labels\[translateToGALabel\(pvTopology\.key\)\]$", "v, ok := node.Labels[translateToGALabel(pvTopology.key)]")],
    "getVolumeLimits": [(r"make\(map\[string\]int64\)", "{}"), (r"int64\(\*d\.Allocatable\.Count\)", "int64(d.Allocatable.Count)")],
    "csiLimitsFilter": [(r"make\(map\[string\]string\)", "{}"), (r"map\[string\]int\{\}", "GoMap()"), (r" /\* (new|existing) pod \*/", ""),
                        (r"^if err := (pl\.filterAttachableVolumes\(.*\)); err != nil \{$", r"if (err := \1) != nil {"),
                        (r"^return fwk\.NewStatus\(fwk\.UnschedulableAndUnresolvable, err\.Error\(\)\)$", 'return ["UnschedulableAndUnresolvable", err.Error()]'),
                        (r"^return fwk\.AsStatus\(err\)$", 'return ["Error", err]'), (r"apierrors\.IsNotFound\(err\)", "err.NotFound"),
                        (r"^delete\(newVolumes, volumeUniqueName\)$", "newVolumes.pop(volumeUniqueName, None)"), (r"^(\w+)\[driverName\]\+\+$", r"\1[driverName] += 1"),
                        (r"^for _, driverName := range newVolumes \{$", "for _, driverName := range newVolumes.values() {"),
                        (r'^"(maxLimits|pod)", .*$', "logger.continued")],
    "checkBoundClaims": [(r"^if errors\.Is\(err, assumecache\.ErrNotFound\) \{$", "if err == ErrNotFound {"), (r"volume\.CheckNodeAffinity\(", "CheckNodeAffinity(")],
    "topologyToMatchedTermCount_update": [(r"^delete\(m, pair\)$", "m.pop(pair, None)")],
    "updateWithAffinityTerms": [(r"\bm\.update\(", "topologyToMatchedTermCount_update(m, ")],
    "updateWithAntiAffinityTerms": [(r"\bm\.update\(", "topologyToMatchedTermCount_update(m, ")],
    "criticalPaths_update": [(r"^p\[1\] = p\[0\]$", "p[1] = gocopy(p[0])")],  # an array element is a struct VALUE: assignment copies
    "calPreFilterState_processNode": [(r"\bc\.matchNodeInclusionPolicies\(", "matchNodeInclusionPolicies(c, ")],
    "calPreFilterState_minima": [(r"^s\.CriticalPaths\[i\]\.update\((.+)\)$", r"criticalPaths_update(s.CriticalPaths[i], \1)")],
    "matchNodeInclusionPolicies": [(r"helper\.DoNotScheduleTaintsFilterFunc\(\)", "DoNotScheduleTaintsFilter_closure"), (r"v1helper\.FindMatchingUntoleratedTaint\(", "FindMatchingUntoleratedTaint("),
                                   (r"v1\.NodeInclusionPolicyHonor", "NodeInclusionPolicyHonor")],
    "RunScorePlugins_weigh": [(r"^err := fmt\.Errorf\(.*$", "err = 'invalid score'"), (r"^errCh\.SendErrorWithCancel\(err, cancel\)$", "errCh.append(err)"),
                              (r"framework\.MinNodeScore", "MinNodeScore"), (r"nodeScoreList\[index\]\.Score", "nodeScoreList[index]"), (r"make\(\[\]framework\.PluginScore, len\(plugins\)\)", "[None] * len(plugins)")],
    "selectHost": [(r"^var h nodeScoreHeap = nodeScoreList$", "h = GoHeap(nodeScoreList)"), (r"^heap\.Init\(&h\)$", "heap.Init(h)"), (r"heap\.Pop\(&h\)\.\(framework\.NodePluginScores\)", "heap.Pop(h)"),
                   (r"make\(\[\]framework\.NodePluginScores, 0, count\)", "[]"), (r"^return \"\", nil, errEmptyPriorityList$", "return '', None, 'empty priorityList'"),
                   (r"^return sortedNodeScoreList\[0\]\.Name, sortedNodeScoreList, nil$", "return sortedNodeScoreList[0].Name, sortedNodeScoreList, None"),
                   (r"^sortedNodeScoreList = sortedNodeScoreList\[:count\]$", "sortedNodeScoreList = sortedNodeScoreList[:count]")],
    "topologyNormalizingWeight": [(r"math\.Log\(", "go_math_log(")],
    "findNodesThatPassFilters_checkNode": [(r"fwk\.Error\b", '"Error"'), (r"^errCh\.SendErrorWithCancel\(status\.AsError\(\), func\(\) \{$", "if errCh.send(status) {"),
                                           (r"errors\.New\((\"[^\"]*\")\)", r"\1"), (r"^\}\)$", "}"),
                                           (r"^length := atomic\.AddInt32\(&feasibleNodesLen, 1\)$", "length := feasibleNodesLen.add(1)"),
                                           (r"^atomic\.AddInt32\(&feasibleNodesLen, -1\)$", "feasibleNodesLen.add(-1)"),
                                           (r"^result\[i\] = &nodeStatus\{node: nodeInfo\.Node\(\)\.Name, status: status\}$", "result[i] = (nodeInfo.Node().Name, status)")],
    "HostPortInfo_sanitize": [(r"\*ip\b", "ip[0]"), (r"\*protocol\b", "protocol[0]"), (r"string\(v1\.ProtocolTCP\)", "ProtocolTCP")],
    "NewProtocolPort": [(r"string\(v1\.ProtocolTCP\)", "ProtocolTCP")],
    "HostPortInfo_CheckConflict": [(r"^h\.sanitize\(&ip, &protocol\)$", "ipp, protop = [ip], [protocol]; HostPortInfo_sanitize(h, ipp, protop); ip, protocol = ipp[0], protop[0]"),
                                   (r"^for _, m := range h \{$", "for _, m := range h.values() {"), (r"^if _, ok := m\[\*pp\]; ok \{$", "if (pp.Protocol, pp.Port) in m {"),
                                   (r"^if _, ok2 := m\[\*pp\]; ok2 \{$", "if (pp.Protocol, pp.Port) in m {"),
                                   (r"^for _, key := range \[\]string\{DefaultBindAllHostIP, ip\} \{$", "for key in [DefaultBindAllHostIP, ip] {")],
    "fitsPorts": [(r"existingPorts\.CheckConflict\(", "HostPortInfo_CheckConflict(existingPorts, "), (r"string\(cp\.Protocol\)", "cp.Protocol")],
    "scoreMap_processTerm": [(r"= make\(map\[string\]int64\)$", "= GoMap()")],
    "scoreMap_processTerms": [(r"^m\.processTerm\(", "scoreMap_processTerm(m, ")],
    "ipa_processExistingPod": [(r"^topoScore\.processTerms\(", "scoreMap_processTerms(topoScore, "), (r"^topoScore\.processTerm\(", "scoreMap_processTerm(topoScore, ")],
    "ipaPreScore_processNode": [(r"= make\(scoreMap\)$", "= GoPtrMap()"), (r"^pl\.processExistingPod\(", "ipa_processExistingPod(pl, "),
                                (r"^topoScores\[atomic\.AddInt32\(&index, 1\)\] = topoScore$", "topoScores.append(topoScore)")],
    "ptsPreScore_initNodes": [(r"v1\.LabelHostname", "LabelHostname"), (r"= new\(int64\)$", "= [0]"), (r"^topoSize\[i\]\+\+$", "topoSize[i] += 1")],
    "ptsPreScore_weights": [(r"v1\.LabelHostname", "LabelHostname")],
    "ptsPreScore_processAllNode": [(r"\bc\.matchNodeInclusionPolicies\(", "matchNodeInclusionPolicies(c, "), (r"^atomic\.AddInt64\(tpCount, int64\(count\)\)$", "tpCount[0] += count")],
    "ptsScore": [(r"v1\.LabelHostname", "LabelHostname"), (r"= \*s\.TopologyValueToPodCounts\[i\]\[tpVal\]$", "= s.TopologyValueToPodCounts[i][tpVal][0]"), (r"math\.Round\(", "go_round(")],
}
# statements about the scheduler's cycle state, not arithmetic: removed before the transliteration (they are still in the recorded Go text)
JOINED = {}
DROP = {
    "volumeZoneFilter": ["logger := klog.FromContext(ctx)", "if len(pod.Spec.Volumes) == 0 {", "return nil", "}", "var podPVTopologies []pvTopology", "state, err := getStateData(cs)",
                         "if err != nil {", "var status *fwk.Status", "podPVTopologies, status = pl.getPVbyPod(logger, pod)", "if !status.IsSuccess() {", "return status", "}", "} else {",
                         "podPVTopologies = state.podPVTopologies", "}"],
    "ptsNormalizeScore": ["s, err := getPreScoreState(cycleState)", "if err != nil {", "return fwk.AsStatus(err)", "}", "if s == nil {", "return nil", "}"],
    "ptsFilter": ["node := nodeInfo.Node()", "s, err := getPreFilterState(cycleState)", "if err != nil {", "return fwk.AsStatus(err)", "}"],
    "ipaFilter": ["state, err := getPreFilterState(cycleState)", "if err != nil {", "return fwk.AsStatus(err)", "}"],
    "ptsScore": ["node := nodeInfo.Node()", "s, err := getPreScoreState(cycleState)", "if err != nil {", "return 0, fwk.AsStatus(err)", "}"],
    "ipaScore": ["node := nodeInfo.Node()", "s, err := getPreScoreState(cycleState)", "if err != nil {", "return 0, fwk.AsStatus(err)", "}"],
    "ipaNormalizeScore": ["s, err := getPreScoreState(cycleState)", "if err != nil {", "return fwk.AsStatus(err)", "}", "if len(s.topologyScore) == 0 {", "return nil", "}"],
}


def cut(rel, start_line, nth=0):
    """The body of the function / closure / block whose first line is `start_line` (its `nth` occurrence): up to the brace that closes it."""
    lines = open(os.path.join(REF, rel)).read().split("\n")
    at = -1
    for _ in range(nth + 1):
        at = lines.index(start_line, at + 1)
    depth, body, opened = 0, [], False
    for ln in lines[at:]:
        depth += ln.count("{") - ln.count("}")
        opened = opened or "{" in ln
        body.append(ln)
        if opened and depth == 0:
            break
    return at + 1, body


def transliterate(name, params, body, int_div, opts=None):
    """Go subset -> Python, one line at a time.  Indentation follows the braces."""
    opts = opts or {}
    out = [f"def {name}({', '.join(params)}):"]
    depth = 1
    if opts.get("oneline"):  # `func ... { return X }`
        return out[0] + "\n    " + expr(re.search(r"\{ (.+) \}$", body[0]).group(1), int_div) + "\n"
    inner = body if opts.get("block") else body[opts.get("header", 1):-1]
    if not opts.get("block") and body[-1].strip() not in ("}", "}, metrics.Score)"):
        raise SystemExit(f"{name}: unexpected closing line {body[-1]!r}")
    named = re.search(r"\) \((\w+) (\[\])?[\w.]+\) \{$", body[0])  # a named result: starts at its zero value, a bare `return` returns it
    if named:
        out.append("    " + named.group(1) + (" = []" if named.group(2) else " = 0"))
    drop = list(DROP.get(name, []))
    while drop:  # the dropped statements are the first non-blank lines of the body, in this order
        while not inner[0].strip() or inner[0].strip().startswith("//"):
            inner = inner[1:]
        if inner[0].strip() != drop[0]:
            raise SystemExit(f"{name}: expected {drop[0]!r}, found {inner[0]!r}")
        inner, drop = inner[1:], drop[1:]
    # a condition continued after && / ||, and a composite literal spread over several lines, become one line each (JOINED counts the
    # absorbed lines for the line-by-line audit)
    merged, k = [], 0
    while k < len(inner):
        ln = inner[k].rstrip()
        while ln.endswith("&&") or ln.endswith("||"):
            k += 1
            ln = ln + " " + inner[k].strip()
            JOINED[name] = JOINED.get(name, 0) + 1
        mlit = re.search(r"(\w+(?:\.\w+)?)\{$", ln)
        if mlit and not ln.endswith("InsufficientResource{") and not ln.startswith(("if ", "for ", "} else", "switch ", "func ")) and " := func(" not in ln and not ln.endswith("func(index int) {"):
            fields = []  # a struct literal spread over several lines -> GoStruct(field=..., ...)
            k += 1
            while inner[k].strip() not in ("}", "})"):
                f = inner[k].strip()
                m = re.fullmatch(r"(\w+):\s+(.+),", f)
                if not m:
                    raise SystemExit(f"{name}: composite literal field {f!r}")
                fields.append(f"{m.group(1)}={m.group(2)}")
                JOINED[name] = JOINED.get(name, 0) + 1
                k += 1
            if inner[k].strip() == "})":
                JOINED[name] = JOINED.get(name, 0) + 1  # the closing line (a bare `}` is a brace like any other to the audit)
            ln = ln[: mlit.start()] + "GoStruct(" + ", ".join(fields) + ")" + (")" if inner[k].strip() == "})" else "")
        if ln.endswith("InsufficientResource{"):
            fields = []
            k += 1
            while inner[k].strip() != "})":
                f = inner[k].strip()
                m = re.fullmatch(r"(\w+):\s+(.+),", f)
                if not m:
                    raise SystemExit(f"{name}: composite literal field {f!r}")
                fields.append(f"{m.group(1)}={m.group(2)}")
                JOINED[name] = JOINED.get(name, 0) + 1
                k += 1
            JOINED[name] = JOINED.get(name, 0) + 1  # the closing line
            ln = ln[: -len("InsufficientResource{")] + "dict(" + ", ".join(fields) + "))"
        merged.append(ln)
        k += 1
    inner = merged
    switches = []  # depth of every open `switch`: its cases become an if / elif chain on _sw
    for raw in inner:
        ln = raw.strip()
        if not ln or ln.startswith("//"):
            continue
        for pat, rep in REWRITE.get(name, []):
            ln = re.sub(pat, rep, ln)
        m = re.fullmatch(r"switch (.+) \{", ln)
        if m:
            out.append("    " * depth + "_sw = " + expr(m.group(1), int_div))
            switches.append([depth, True])
            depth += 1
            continue
        if switches and (re.fullmatch(r"case (.+):", ln) or ln == "default:"):
            d, first = switches[-1]
            if ln == "default:":
                out.append("    " * d + "else:")
            else:
                vals = ", ".join(expr(v.strip(), int_div) for v in ln[5:-1].split(","))
                out.append("    " * d + ("if" if first else "elif") + f" _sw in ({vals},):")
                switches[-1][1] = False
            depth = d + 1
            continue
        # closing braces (with else) first
        if ln.startswith("}"):
            depth -= 1
            if switches and depth == switches[-1][0]:
                switches.pop()
            ln = ln[1:].strip()
            if not ln:
                continue
            if ln.startswith("else if ") and ln.endswith("{"):
                ln = "elif " + ln[len("else if "):-1].strip() + ":"
            elif ln == "else {":
                ln = "else:"
            else:
                raise SystemExit(f"{name}: cannot transliterate {raw!r}")
            out.append("    " * depth + expr(ln, int_div))
            depth += 1
            continue
        opens = ln.endswith("{")
        if opens:
            ln = ln[:-1].strip()
            m = re.fullmatch(r"for (\w+) := range ([\w.]+)", ln)
            m2 = re.fullmatch(r"for _, (\w+) := range ([\w.]+(?:\(\))?)", ln)
            m3 = re.fullmatch(r"for (\w+), (\w+) := range ([\w.]+)", ln)
            m4 = re.fullmatch(r"for (\w+), (\w+) := range ([\w.]+\.ScalarResources)", ln)
            mmap = re.fullmatch(r"for (\w+), (\w+) := range (other|oScores|s\.topologyScore|attachedVolumes|volumeAttachments|newVolumeCount)", ln)  # the score maps of InterPodAffinity; the CSI volume maps
            if m4 or mmap:  # a map: Go's order is random, sorted here
                m4 = m4 or mmap
                ln = f"for {m4.group(1)}, {m4.group(2)} in sorted({m4.group(3)}.items()):"
            elif m3 and m3.group(1) != "_":
                ln = f"for {m3.group(1)}, {m3.group(2)} in enumerate({m3.group(3)}):"
            elif m:
                ln = f"for {m.group(1)} in range(len({m.group(2)})):"
            elif m2:
                ln = f"for {m2.group(1)} in {m2.group(2)}:"
            elif re.fullmatch(r"if (\w+), (\w+) := (.+)\[([\w.]+)\]; \2", ln):  # if with an init statement: the lookup, then the test
                mi = re.fullmatch(r"if (\w+), (\w+) := (.+)\[([\w.]+)\]; \2", ln)
                out.append("    " * depth + f"{mi.group(2)} = {mi.group(4)} in {mi.group(3)}")
                out.append("    " * depth + f"{mi.group(1)} = {mi.group(3)}.get({mi.group(4)}, \"\")")
                ln = f"if {mi.group(2)}:"
            elif re.fullmatch(r"for (\w+) := 0; \1 < len\((\w+)\); \1\+\+", ln):  # the counting loop
                mc = re.fullmatch(r"for (\w+) := 0; \1 < len\((\w+)\); \1\+\+", ln)
                ln = f"for {mc.group(1)} in range(len({mc.group(2)})):"
            elif re.fullmatch(r"for (\w+), (\w+) := range (s\.TpValueToMatchNum\[i\])", ln):  # a map: Go's order is random, sorted here
                mc = re.fullmatch(r"for (\w+), (\w+) := range (s\.TpValueToMatchNum\[i\])", ln)
                ln = f"for {mc.group(1)}, {mc.group(2)} in sorted({mc.group(3)}.items()):"
            elif re.fullmatch(r"if _, (\w+) := ([\w.]+)\[([\w.]+)\]; !\1", ln):  # presence test with an init statement
                mc = re.fullmatch(r"if _, (\w+) := ([\w.]+)\[([\w.]+)\]; !\1", ln)
                out.append("    " * depth + f"{mc.group(1)} = {mc.group(3)} in {mc.group(2)}")
                ln = f"if not {mc.group(1)}:"
            elif re.fullmatch(r"if (\w+), _ := (.+); !\1", ln):  # `if match, _ := x.Match(node); !match {`
                mc = re.fullmatch(r"if (\w+), _ := (.+); !\1", ln)
                out.append("    " * depth + f"{mc.group(1)}, _ = {expr(mc.group(2), int_div)}")
                ln = f"if not {mc.group(1)}:"
            elif re.fullmatch(r"if _, (\w+) := (.+); \1", ln):  # `if _, untolerated := f(...); untolerated {`
                mc = re.fullmatch(r"if _, (\w+) := (.+); \1", ln)
                out.append("    " * depth + f"_, {mc.group(1)} = {expr(mc.group(2), int_div)}")
                ln = f"if {mc.group(1)}:"
            elif re.fullmatch(r"for (\w+) := (.+); ; \1 = \2", ln):  # `for x := f(); ; x = f() {`: f() before every iteration
                mc = re.fullmatch(r"for (\w+) := (.+); ; \1 = \2", ln)
                out.append("    " * depth + "while True:")
                depth += 1
                out.append("    " * depth + f"{mc.group(1)} = {expr(mc.group(2), int_div)}")
                continue
            elif re.fullmatch(r"for (\w+), (\w+) := range ([\w.()]+\.Labels)", ln):
                ml = re.fullmatch(r"for (\w+), (\w+) := range ([\w.()]+\.Labels)", ln)
                ln = f"for {ml.group(1)}, {ml.group(2)} in sorted({ml.group(3)}.items()):"
            elif ln.startswith("if "):
                ln = "if " + ln[3:] + ":"
            elif re.fullmatch(r"for \w+ in \[[\w., ]+\]", ln):  # (a REWRITE put a Go range over a slice literal into this form)
                ln = ln + ":"
            else:
                raise SystemExit(f"{name}: cannot transliterate {raw!r}")
        else:
            m = re.fullmatch(r"var (\w+), (\w+) (int64|float64)", ln)
            if m:
                ln = f"{m.group(1)} = {m.group(2)} = 0"
            elif re.fullmatch(r"var (\w+) \[\]float64", ln):
                ln = re.sub(r"var (\w+) \[\]float64", r"\1 = []", ln)
            elif re.fullmatch(r"var (\w+) int64 = (.+)", ln):
                ln = re.sub(r"var (\w+) int64 = (.+)", r"\1 = \2", ln)
            elif re.fullmatch(r"var (\w+) (int64|int32|float64)", ln):
                ln = re.sub(r"var (\w+) (int64|int32|float64)", r"\1 = 0", ln)
            if ln.startswith("klog.") or ln.startswith("logger"):
                ln = "pass"
            ln = re.sub(r"^var (\w+) string$", r'\1 = ""', ln)
            ln = re.sub(r"make\(\[\]\w+, 0, \d+\)", "[]", ln)
            ln = re.sub(r"make\(\[\]\w+, 0, len\(\w+\)\)", "[]", ln)
            mk = re.fullmatch(r"_, (\w+) := (.+)\[(\w+)\]", ln)
            if mk:  # presence only
                ln = f"{mk.group(1)} = {mk.group(3)} in {mk.group(2)}"
            ml = re.fullmatch(r"(\w+), (\w+) := ls\.Lookup\((.+)\)", ln)
            mp = re.fullmatch(r"(\w+), err :?= strconv\.ParseInt\((.+), 10, 64\)", ln)
            if ml:  # Labels.Lookup: (value, present)
                out.append("    " * depth + f"{ml.group(2)} = {ml.group(3)} in ls")
                ln = f"{ml.group(1)} = ls.get({ml.group(3)}, \"\")"
            elif mp:
                ln = f"{mp.group(1)}, err = parse_int({mp.group(2)})"
            m = re.fullmatch(r"(\w+), ok := ([\w.]+)\[(.+)\]", ln)
            m2 = re.fullmatch(r"(\w+), _ = (\w+)\[(.+)\]", ln)
            if m:   # the two-value map lookup: the zero value ("") when the key is absent
                out.append("    " * depth + f"ok = {expr(m.group(3), int_div)} in {m.group(2)}")
                ln = f"{m.group(1)} = {m.group(2)}.get({m.group(3)}, \"\")"
            elif m2:
                ln = f"{m2.group(1)} = {m2.group(2)}.get({m2.group(3)}, \"\")"
            ln = ln.replace(":=", "=")
            m = re.fullmatch(r"(\w+) = append\((\w+), (.+)\)", ln)
            if m and m.group(1) == m.group(2):
                ln = f"{m.group(1)} = {m.group(1)} + [{m.group(3)}]"
            if ln == "return nil":
                ln = "return None"
            if ln == "return" and named:
                ln = "return " + named.group(1)
            ln = re.sub(r"^([\w.]+)\+\+$", r"\1 += 1", ln)
        out.append("    " * depth + expr(ln, int_div))
        if opens:
            depth += 1
    return "\n".join(out) + "\n"


def expr(ln, int_div):
    ln = ln.replace("framework.MaxNodeScore", "MaxNodeScore").replace("math.Abs(", "abs(").replace("math.Sqrt(", "math.sqrt(")
    ln = re.sub(r"\bfloat64\(", "float(", ln)
    ln = re.sub(r"\bint64\(", "goint(", ln)
    ln = re.sub(r"\bint32\((\d+)\)", r"\1", ln)
    ln = ln.replace("scores[i].Score", "scores[i]").replace("resources[i].Weight", "resources[i]")
    ln = ln.replace("s.IgnoredNodes.Has(score.Name)", "ignored[i]").replace("score.Score", "score").replace("math.MaxInt64", "MaxInt64").replace("math.MinInt64", "MinInt64")
    ln = re.sub(r"^(\s*)(\w+) = float\(0\)$", r"\1\2 = 0.0", ln)
    ln = ln.replace("percentageOfNodesToScore != nil", "percentageOfNodesToScore is not None").replace("*percentageOfNodesToScore", "percentageOfNodesToScore")
    ln = ln.replace("true", "True").replace("false", "False") if re.search(r"\b(true|false)\b", ln) else ln
    ln = re.sub(r"strings\.LastIndex\((\w+), (\"[^\"]*\")\)", r"\1.rfind(\2)", ln)
    ln = re.sub(r"\br\.hasValue\((\w+)\)", r"requirementHasValue(r, \1)", ln)
    ln = re.sub(r"\bls\.Has\(([\w.]+)\)", r"(\1 in ls)", ln)
    ln = re.sub(r"\bselection\.(\w+)", r"SEL_\1", ln)
    ln = re.sub(r"\b(\w+)\[i\]\.ToleratesTaint\((\w+)\)", r"ToleratesTaint(\1[i], \2)", ln)
    ln = ln.replace("v1helper.TolerationsTolerateTaint(", "TolerationsTolerateTaint(").replace("[]v1.Taint{}", "[]").replace("v1.Taint{}", "None")
    ln = ln.replace("s.minMatchNum(", "preFilterState_minMatchNum(s, ").replace("labels.Set(", "(")
    ln = re.sub(r"topologyPair\{key: ([\w.]+), value: (\w+)\}", r"(\1, \2)", ln)
    ln = re.sub(r"fwk\.NewStatus\(fwk\.(\w+), (\w+)\)", r'["\1", \2]', ln)
    ln = re.sub(r"\bv1\.TaintEffect(\w+)", r"TaintEffect\1", ln)
    ln = re.sub(r"\bv1\.Resource(Pods|CPU|Memory|EphemeralStorage)\b", r"Resource\1", ln).replace("v1helper.IsExtendedResourceName(", "IsExtendedResourceName(")
    ln = re.sub(r'fmt\.Sprintf\("([^"%]*)%v", (\w+)\)', r'("\1%s" % \2)', ln)
    ln = re.sub(r'strings\.Split\(string\((\w+)\), ("[^"]*")\)', r"\1.split(\2)", ln)
    ln = re.sub(r"\bstring\((\w+)\)", r"str(\1)", ln)
    ln = ln.replace(" && ", " and ").replace(" || ", " or ")
    ln = re.sub(r"(?<![&\w])&(?=\w)", "", ln)  # &taint: the address of the loop variable, read only
    ln = re.sub(r"!(?=[\w(])", "not ", ln)  # (logical not; != is left alone)
    ln = re.sub(r"\bnil\b", "None", ln)
    ln = ln.replace("v1.LabelFailureDomainBetaZone", "LabelFailureDomainBetaZone").replace("v1.LabelTopologyZone", "LabelTopologyZone")
    ln = ln.replace("v1.LabelFailureDomainBetaRegion", "LabelFailureDomainBetaRegion").replace("v1.LabelTopologyRegion", "LabelTopologyRegion")
    if int_div and "/" in ln:
        # a / b between integers: Go truncates toward zero.  Only the shapes that occur: `X / name` and `X / number`, at the top level of
        # a statement `lhs = A / B`, `return A / B` or `lhs = A - B/C`
        m = re.fullmatch(r"(\s*(?:return |[\w\[\]]+ = ))(.+) / (\w+|\([^()]*\))", ln)
        m2 = re.fullmatch(r"(\s*\w+ = )(\w+) - (\w+)/(\d+)", ln)
        if m2:
            ln = f"{m2.group(1)}{m2.group(2)} - godiv({m2.group(3)}, {m2.group(4)})"
        elif m:
            ln = f"{m.group(1)}godiv({m.group(2)}, {m.group(3)})"
        else:
            raise SystemExit(f"integer division of an unexpected shape: {ln!r}")
    return ln


SELECTION = {"DoesNotExist": "!", "Equals": "=", "DoubleEquals": "==", "In": "in", "NotEquals": "!=", "NotIn": "notin", "Exists": "exists", "GreaterThan": "gt",
             "LessThan": "lt"}  # apimachinery/pkg/selection/operator.go:23-33 (checked against the file in build())


RESOURCE_NAMES = {"Pods": "pods", "CPU": "cpu", "Memory": "memory", "EphemeralStorage": "ephemeral-storage"}  # api/core/v1/types.go (checked in build())
TAINT_EFFECTS = {"NoSchedule": "NoSchedule", "PreferNoSchedule": "PreferNoSchedule", "NoExecute": "NoExecute"}  # api/core/v1/types.go (checked in build())


class GoMap(dict):
    """map[K]int64: a missing key reads as 0."""
    def __missing__(self, k):
        return 0


class GoResource:
    """framework.Resource behind its getters (S/framework/types.go)."""
    def __init__(self, cpu=0, mem=0, eph=0, pods=0, scalars=None):
        self.cpu, self.mem, self.eph, self.pods, self.scalars = cpu, mem, eph, pods, GoMap(scalars or {})

    def GetMilliCPU(self): return self.cpu
    def GetMemory(self): return self.mem
    def GetEphemeralStorage(self): return self.eph
    def GetAllowedPodNumber(self): return self.pods
    def GetScalarResources(self): return self.scalars


class GoNodeInfo:
    def __init__(self, alloc, requested, n_pods):
        self.alloc, self.requested, self.pods = alloc, requested, [None] * n_pods

    def GetAllocatable(self): return self.alloc
    def GetRequested(self): return self.requested
    def GetPods(self): return self.pods


class GoPodRequest:
    """preFilterState (fit.go:100-108): the embedded Resource's fields + the DRA mapping (nil without the feature)."""
    def __init__(self, cpu, mem, eph, scalars):
        self.MilliCPU, self.Memory, self.EphemeralStorage, self.ScalarResources, self.resourceToDeviceClass = cpu, mem, eph, GoMap(scalars), None

    def GetEphemeralStorage(self): return self.EphemeralStorage


class GoNilSet:
    """a nil sets.Set[string]."""
    def Len(self): return 0
    def Has(self, k): return False


class GoStruct(types.SimpleNamespace):
    """A Go struct value: fields that were not set read as the zero value 0 (the numeric ones are the only ones read unset)."""
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return 0


def gocopy(x):
    import copy
    return copy.copy(x)


class GoLabels(dict):
    """map[string]string: a missing key reads as ""."""
    def __missing__(self, k):
        return ""


class GoPtrMap(dict):
    """map[string]*int64: a missing key reads as nil."""
    def __missing__(self, k):
        return None


class GoSet(set):
    """sets.Set[string]."""
    def Insert(self, k): self.add(k)
    def Has(self, k): return k in self


def go_round(x):
    """math.Round: half away from zero."""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


class GoHeap(list):
    """nodeScoreHeap: a slice behind heap.Interface -- Len / Less / Swap / Push / Pop as schedule_one.go:949-963 has them (Less is the transliterated one)."""
    less = None

    def Len(self): return len(self)
    def Less(self, i, j): return GoHeap.less(self, i, j)
    def Swap(self, i, j): self[i], self[j] = self[j], self[i]
    def Pop(self): return list.pop(self)


class GoContainerHeap:
    """container/heap of Go's standard library (not part of the reference tree): Init and Pop with the documented sift-down."""
    @staticmethod
    def _down(h, i0, n):
        i = i0
        while True:
            j1 = 2 * i + 1
            if j1 >= n or j1 < 0:
                break
            j = j1
            j2 = j1 + 1
            if j2 < n and h.Less(j2, j1):
                j = j2
            if not h.Less(j, i):
                break
            h.Swap(i, j)
            i = j
        return i > i0

    @staticmethod
    def Init(h):
        n = h.Len()
        for i in range(n // 2 - 1, -1, -1):
            GoContainerHeap._down(h, i, n)

    @staticmethod
    def Pop(h):
        n = h.Len() - 1
        h.Swap(0, n)
        GoContainerHeap._down(h, 0, n)
        return h.Pop()


class ScriptedRand:
    """math/rand behind selectHost: Intn answers from a script (0 at the scripted call, 1 otherwise: 'keep the candidate')."""
    def __init__(self, zero_at):
        self.calls, self.zero_at = 0, zero_at

    def Intn(self, n):
        self.calls += 1
        return 0 if self.calls == self.zero_at else 1


def go_math_log(x):
    """Go's math.Log, pure-Go path (src/math/log.go = FreeBSD e_log.c; not part of the reference tree): IEEE double +, -, *, / and frexp only, so this
    restatement is bit-identical to it -- Python floats are IEEE doubles, and CPython fuses nothing."""
    Ln2Hi, Ln2Lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    L1, L2, L3, L4 = 6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01
    L5, L6, L7 = 1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01
    f1, ki = math.frexp(x)
    if f1 < math.sqrt(2) / 2:
        f1 *= 2
        ki -= 1
    f = f1 - 1
    k = float(ki)
    s_ = f / (2 + f)
    s2 = s_ * s_
    s4 = s2 * s2
    t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)))
    t2 = s4 * (L2 + s4 * (L4 + s4 * L6))
    R = t1 + t2
    hfsq = 0.5 * f * f
    return k * Ln2Hi - ((hfsq - (s_ * (hfsq + R) + k * Ln2Lo)) - f)


def parse_int(text):
    """strconv.ParseInt(text, 10, 64): (value, nil) or (0, error).  Go's standard library is not part of the reference tree; this is its
    documented contract: an optional sign, decimal digits, inside int64."""
    if re.fullmatch(r"[+-]?[0-9]+", text) and -(1 << 63) <= int(text) < (1 << 63):
        return int(text), None
    return 0, "error"



# ---- round 6: the volume plugins' Filters on Kubernetes objects (dicts as `kubectl get -o json` gives them) -----------------------------------------
class GoObj(types.SimpleNamespace):
    """A Go struct with exactly the fields the harness sets (a pointer field that is nil is None)."""


ErrNotFound = "assumecache: object not found"  # the sentinel checkBoundClaims compares with (errors.Is)


class GoNotFound:
    """apierrors.NewNotFound(resource, name): Error() as apimachinery formats it."""
    NotFound = True

    def __init__(self, kind, name):
        self.msg = f'{kind} "{name}" not found'

    def Error(self):
        return self.msg


def go_volume(v):
    g, a, i, r, c = v.get("gcePersistentDisk"), v.get("awsElasticBlockStore"), v.get("iscsi"), v.get("rbd"), v.get("persistentVolumeClaim")
    return GoObj(Name=v.get("name", ""),
                 GCEPersistentDisk=None if g is None else GoObj(PDName=g.get("pdName", ""), ReadOnly=bool(g.get("readOnly"))),
                 AWSElasticBlockStore=None if a is None else GoObj(VolumeID=a.get("volumeID", "")),
                 ISCSI=None if i is None else GoObj(IQN=i.get("iqn", ""), ReadOnly=bool(i.get("readOnly"))),
                 RBD=None if r is None else GoObj(CephMonitors=list(r.get("monitors") or []), RBDPool=r.get("pool") or "", RBDImage=r.get("image", ""), ReadOnly=bool(r.get("readOnly"))),
                 PersistentVolumeClaim=None if c is None else GoObj(ClaimName=c.get("claimName", "")), Ephemeral=None)


def go_pod(p):
    md = p.get("metadata") or {}
    return GoObj(Name=md.get("name", ""), Namespace=md.get("namespace") or "default", Spec=GoObj(Volumes=[go_volume(v) for v in (p.get("spec") or {}).get("volumes") or []]))


class GoPodInfo:
    def __init__(self, pod): self.pod = pod
    def GetPod(self): return self.pod


class GoVolNodeInfo:
    def __init__(self, name, labels, pods):
        self.node, self.pods = GoObj(Name=name, Labels=dict(labels)), [GoPodInfo(p) for p in pods]

    def Node(self): return self.node
    def GetPods(self): return self.pods


def label_zones_to_set(value):
    """volumehelpers.LabelZonesToSet (cloud-provider/volume/helpers): "a__b" -> {a, b}; an empty element is an error.  Not part of the cut: the
    reference calls it while PreFilter collects the topologies (volume_zone.go:380-398, restated in pv_topologies below)."""
    out = GoSet()
    for z in value.split("__"):
        z = z.strip()
        if not z:
            return None
        out.add(z)
    return out


def pv_topologies(env, pv):
    out = []
    labels = (pv.get("metadata") or {}).get("labels") or {}
    for key in env["topologyLabels"]:
        if key in labels:
            zs = label_zones_to_set(labels[key])
            if zs is None:
                continue
            out.append(GoObj(pvName=pv["metadata"]["name"], key=key, values=zs))
    return out


NODE_SELECTOR_OPS = {"In": "in", "NotIn": "notin", "Exists": "exists", "DoesNotExist": "!", "Gt": "gt", "Lt": "lt"}  # nodeaffinity.go nodeSelectorRequirementsAsSelector


def check_node_affinity(env, pv, node_labels):
    """storagehelpers.CheckNodeAffinity (component-helpers/storage/volume/helpers.go:68-84) over corev1.MatchNodeSelectorTerms: the terms are ORed, a
    term's matchExpressions ANDed (an empty term matches nothing); each expression is the transliterated labels.Requirement.Matches."""
    req = ((pv.get("spec") or {}).get("nodeAffinity") or {}).get("required")
    if (pv.get("spec") or {}).get("nodeAffinity") is None or req is None:
        return None
    for term in req.get("nodeSelectorTerms") or []:
        exprs = term.get("matchExpressions") or []
        if not exprs:
            continue
        if all(env["requirementMatches"](types.SimpleNamespace(key=r["key"], operator=NODE_SELECTOR_OPS[r["operator"]], strValues=[str(x) for x in r.get("values") or []]), node_labels)
               for r in exprs):
            return None
    return "no matching NodeSelectorTerms"


class CsiHarness:
    """What CSILimits.Filter reaches through pl.*: the listers, filterAttachableVolumes / getCSIDriverInfo(+FromSC) (csi.go:345-412, 455-545; in-tree
    volumes that would be migrated do not occur in the generated worlds), getNodeVolumeAttachmentInfo (:588-618)."""
    def __init__(self, pvcs, pvs, classes, csinodes, vas):
        self.pvcs, self.pvs, self.classes, self.csinodes, self.vas = pvcs, pvs, classes, csinodes, vas
        self.csiNodeLister = types.SimpleNamespace(Get=self._csinode)

    def _csinode(self, name):
        o = self.csinodes.get(name)
        if o is None:
            return None, GoNotFound("csinode.storage.k8s.io", name)
        drivers = [GoObj(Name=d.get("name", ""), Allocatable=None if d.get("allocatable") is None else GoObj(Count=d["allocatable"].get("count")))
                   for d in (o.get("spec") or {}).get("drivers") or []]
        return GoObj(Name=name, Spec=GoObj(Drivers=drivers)), None

    def _driver_info(self, pvc):
        spec, md = pvc.get("spec") or {}, pvc.get("metadata") or {}
        pv = self.pvs.get(spec.get("volumeName") or "") if spec.get("volumeName") else None
        if pv is None:  # getCSIDriverInfoFromSC
            ann = md.get("annotations") or {}
            sc = ann["volume.beta.kubernetes.io/storage-class"] if "volume.beta.kubernetes.io/storage-class" in ann else (spec.get("storageClassName") or "")
            cls = self.classes.get(sc) if sc else None
            if cls is None:
                return "", ""
            return cls.get("provisioner") or "", f'RANDOMPREFIX-{md.get("namespace") or "default"}/{md.get("name", "")}'
        csi = (pv.get("spec") or {}).get("csi")
        if csi is None:
            return "", ""
        return csi.get("driver") or "", csi.get("volumeHandle") or ""

    def filterAttachableVolumes(self, logger, pod, csiNode, newPod, result):
        for vol in pod.Spec.Volumes:
            if vol.PersistentVolumeClaim is None:
                continue  # (inline volumes: only migratable in-tree ones would count)
            name = vol.PersistentVolumeClaim.ClaimName
            if name == "":
                return "PersistentVolumeClaim had no name"
            pvc = self.pvcs.get((pod.Namespace, name))
            if pvc is None:
                if newPod:
                    return GoNotFound("persistentvolumeclaim", name)
                continue
            driver, handle = self._driver_info(pvc)
            if driver == "" or handle == "":
                continue
            result[f"{driver}/{handle}"] = driver
        return None

    def getNodeVolumeAttachmentInfo(self, logger, nodeName):
        out = {}
        for va in self.vas:
            sp = va.get("spec") or {}
            if sp.get("nodeName") != nodeName or not sp.get("attacher"):
                continue
            pvn = (sp.get("source") or {}).get("persistentVolumeName")
            pv = self.pvs.get(pvn) if pvn is not None else None
            csi = ((pv or {}).get("spec") or {}).get("csi")
            if pv is None or csi is None:
                continue
            out[f'{sp["attacher"]}/{csi.get("volumeHandle") or ""}'] = sp["attacher"]
        return out, None


def volume_filter_rows(env, rnd):
    """Four families of small worlds, one per plugin; per node what the plugin's transliterated Filter says."""
    BIND_DONE = "pv.kubernetes.io/bind-completed"
    zone, zone_b, region = PINS["label.zone"], PINS["label.zone_beta"], PINS["label.region"]
    drivers = ["ebs.csi.aws.com", "pd.csi.storage.gke.io"]

    def mk_nodes():
        n = rnd.randint(2, 9)
        out = []
        for i in range(n):
            labels = {"kubernetes.io/hostname": f"n{i}"}
            if rnd.random() < 0.8:
                labels[zone if rnd.random() < 0.6 else zone_b] = f"z{rnd.randint(0, 2)}"
            if rnd.random() < 0.3:
                labels[region] = f"r{rnd.randint(0, 1)}"
            out.append({"name": f"n{i}", "labels": labels})
        return out

    def disk_volumes(k_max):
        vols = []
        for j in range(rnd.randint(0, k_max)):
            r, name = rnd.random(), f"v{j}"
            if r < 0.3:
                vols.append({"name": name, "gcePersistentDisk": {"pdName": f"disk-{rnd.randint(0, 2)}", **({"readOnly": rnd.random() < 0.5} if rnd.random() < 0.8 else {})}})
            elif r < 0.5:
                vols.append({"name": name, "awsElasticBlockStore": {"volumeID": f"vol-{rnd.randint(0, 2)}"}})
            elif r < 0.7:
                vols.append({"name": name, "rbd": {"monitors": [f"m{rnd.randint(0, 3)}" for _ in range(rnd.randint(0, 3))], **({"pool": rnd.choice(["p", "q"])} if rnd.random() < 0.7 else {}),
                                                   "image": f"i{rnd.randint(0, 1)}", "readOnly": rnd.random() < 0.5}})
            elif r < 0.85:
                vols.append({"name": name, "iscsi": {"iqn": f"iqn-{rnd.randint(0, 1)}", "targetPortal": "p", "lun": 0, "readOnly": rnd.random() < 0.5}})
            else:
                vols.append({"name": name, "emptyDir": {}})
        return vols

    def claim(name, pv_name="", cls=None, ns="default"):
        md = {"name": name, "namespace": ns}
        spec = {"accessModes": ["ReadWriteOnce"]}
        if pv_name:
            md["annotations"] = {BIND_DONE: "yes"}
            spec["volumeName"] = pv_name
        if cls is not None:
            spec["storageClassName"] = cls
        return {"apiVersion": "v1", "kind": "PersistentVolumeClaim", "metadata": md, "spec": spec, "status": {"phase": "Bound" if pv_name else "Pending"}}

    def pod(name, node_name, volumes, ns="default"):
        return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": ns}, "spec": {"nodeName": node_name, "volumes": volumes}}

    def infos(nodes, pods):
        return [GoVolNodeInfo(nd["name"], nd["labels"], [go_pod(p) for p in pods if p["spec"]["nodeName"] == nd["name"]]) for nd in nodes]

    rows = {"volumeRestrictions": [], "volumeZone": [], "csiLimits": [], "boundClaims": []}
    for _ in range(160):
        nodes = mk_nodes()
        pods = [pod(f"p{k}", rnd.choice(nodes)["name"], disk_volumes(3)) for k in range(rnd.randint(0, 8))]
        tmpl = pod("sim", "", disk_volumes(3))
        gp = go_pod(tmpl)
        exp = [0 if env["satisfyVolumeConflicts"](gp, ni) else 1 for ni in infos(nodes, pods)]
        own = GoVolNodeInfo("x", {}, [gp])  # the template's clone on the same node
        rows["volumeRestrictions"].append({"nodes": nodes, "pods": pods, "volumes": tmpl["spec"]["volumes"], "conflict": exp, "exclusive": not env["satisfyVolumeConflicts"](gp, own)})
    for _ in range(160):
        nodes = mk_nodes()
        pvs = []
        for k in range(rnd.randint(1, 5)):
            labels = {}
            for key in (zone, zone_b, region, PINS["label.region_beta"]):
                if rnd.random() < 0.3:
                    labels[key] = rnd.choice(["z0", "z1", "z0__z2", "z1__z2__z0", "r0", "r1", "z0__", " z1 ", ""])
            pvs.append({"apiVersion": "v1", "kind": "PersistentVolume", "metadata": {"name": f"pv-{k}", "labels": labels},
                        "spec": {"capacity": {"storage": "1Gi"}, "csi": {"driver": drivers[0], "volumeHandle": f"h-{k}"}}})
        claims = [claim(f"c{k}", f"pv-{rnd.randint(0, len(pvs) - 1)}") for k in range(rnd.randint(1, 4))]
        vols = [{"name": f"v{j}", "persistentVolumeClaim": {"claimName": c["metadata"]["name"]}} for j, c in enumerate(claims) if rnd.random() < 0.8] or \
               [{"name": "v0", "persistentVolumeClaim": {"claimName": claims[0]["metadata"]["name"]}}]
        by = {o["metadata"]["name"]: o for o in pvs}
        cl = {c["metadata"]["name"]: c for c in claims}
        topo = []
        for v in vols:
            topo += pv_topologies(env, by[cl[v["persistentVolumeClaim"]["claimName"]]["spec"]["volumeName"]])
        gp = go_pod(pod("sim", "", vols))
        exp = [0 if env["volumeZoneFilter"](topo, gp, ni) is None else 1 for ni in infos(nodes, [])]
        rows["volumeZone"].append({"nodes": nodes, "objs": pvs + claims, "volumes": vols, "reject": exp})
    for _ in range(200):
        nodes = mk_nodes()
        classes = [{"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "csi-wait"}, "provisioner": drivers[1], "volumeBindingMode": "WaitForFirstConsumer"},
                   {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "local"}, "provisioner": "kubernetes.io/no-provisioner", "volumeBindingMode": "WaitForFirstConsumer"}]
        pvs = []
        for k in range(rnd.randint(2, 8)):
            spec = {"capacity": {"storage": "1Gi"}}
            if rnd.random() < 0.85:
                spec["csi"] = {"driver": rnd.choice(drivers), "volumeHandle": f"h-{rnd.randint(0, 5)}"}  # (handles repeat: one volume behind two PVs counts once)
            else:
                spec["local"] = {"path": "/mnt"}
            pvs.append({"apiVersion": "v1", "kind": "PersistentVolume", "metadata": {"name": f"pv-{k}", "labels": {}}, "spec": spec})
        claims = []
        for k in range(rnd.randint(2, 9)):
            r = rnd.random()
            ns = "other" if rnd.random() < 0.15 else "default"
            if r < 0.65:
                claims.append(claim(f"c{k}", f"pv-{rnd.randint(0, len(pvs))}", ns=ns))  # (pv-<len> does not exist: the class decides, and there is none)
            elif r < 0.85:
                claims.append(claim(f"c{k}", cls=rnd.choice(["csi-wait", "local", "gone"]), ns=ns))
            else:
                claims.append(claim(f"c{k}", ns=ns))
        mine = [c for c in claims if c["metadata"]["namespace"] == "default"] or [claim("c-own", "pv-0")]
        if mine[0] not in claims:
            claims.append(mine[0])

        def claim_vols(k_max, ns, known_only):
            out = []
            for j in range(rnd.randint(0, k_max)):
                pool = [c["metadata"]["name"] for c in claims if c["metadata"]["namespace"] == ns]
                if not known_only:
                    pool.append("missing")
                if pool:
                    out.append({"name": f"v{j}", "persistentVolumeClaim": {"claimName": rnd.choice(pool)}})
            return out
        pods = []
        for k in range(rnd.randint(0, 10)):
            ns = "other" if rnd.random() < 0.2 else "default"
            pods.append(pod(f"p{k}", rnd.choice(nodes)["name"], claim_vols(3, ns, False), ns))
        vols = claim_vols(4, "default", True) or [{"name": "v0", "persistentVolumeClaim": {"claimName": mine[0]["metadata"]["name"]}}]
        csinodes = []
        for nd in nodes:
            if rnd.random() < 0.75:
                drv = []
                for d in drivers:
                    if rnd.random() < 0.7:
                        e = {"name": d, "nodeID": nd["name"]}
                        if rnd.random() < 0.85:
                            e["allocatable"] = {"count": rnd.randint(0, 3)} if rnd.random() < 0.9 else {}
                        drv.append(e)
                csinodes.append({"apiVersion": "storage.k8s.io/v1", "kind": "CSINode", "metadata": {"name": nd["name"]}, "spec": {"drivers": drv}})
        vas = [{"apiVersion": "storage.k8s.io/v1", "kind": "VolumeAttachment", "metadata": {"name": f"va{k}"},
                "spec": {"attacher": rnd.choice(drivers), "nodeName": rnd.choice(nodes)["name"], "source": {"persistentVolumeName": f"pv-{rnd.randint(0, len(pvs))}"}}}
               for k in range(rnd.randint(0, 5))]
        h = CsiHarness({(c["metadata"]["namespace"], c["metadata"]["name"]): c for c in claims}, {o["metadata"]["name"]: o for o in pvs}, {o["metadata"]["name"]: o for o in classes},
                       {o["metadata"]["name"]: o for o in csinodes}, vas)
        gp = go_pod(pod("sim", "", vols))
        exp = []
        for ni in infos(nodes, pods):
            st = env["csiLimitsFilter"](h, gp, ni)
            assert st is None or st[0] == "Unschedulable", st
            exp.append(0 if st is None else 1)
        rows["csiLimits"].append({"nodes": nodes, "pods": pods, "objs": classes + pvs + claims + csinodes + vas, "volumes": vols, "reject": exp})
    for _ in range(160):
        nodes = mk_nodes()
        pvs = []
        for k in range(rnd.randint(1, 5)):
            spec = {"capacity": {"storage": "1Gi"}, "csi": {"driver": drivers[0], "volumeHandle": f"h-{k}"}}
            r = rnd.random()
            if r < 0.6:
                terms = []
                for _t in range(rnd.randint(0, 2)):
                    ex = []
                    for _e in range(rnd.randint(0, 2)):
                        op = rnd.choice(["In", "NotIn", "Exists", "DoesNotExist"])
                        key = rnd.choice(["kubernetes.io/hostname", zone, "absent"])
                        e = {"key": key, "operator": op}
                        if op in ("In", "NotIn"):
                            e["values"] = [rnd.choice([f"n{rnd.randint(0, 8)}", f"z{rnd.randint(0, 2)}"]) for _v in range(rnd.randint(1, 3))]
                        ex.append(e)
                    terms.append({"matchExpressions": ex} if ex or rnd.random() < 0.5 else {})
                spec["nodeAffinity"] = {"required": {"nodeSelectorTerms": terms}}
            elif r < 0.7:
                spec["nodeAffinity"] = {}
            pvs.append({"apiVersion": "v1", "kind": "PersistentVolume", "metadata": {"name": f"pv-{k}", "labels": {}}, "spec": spec})
        claims = [claim(f"c{k}", f"pv-{rnd.randint(0, len(pvs) - (0 if rnd.random() < 0.25 else 1))}") for k in range(rnd.randint(1, 4))]
        vols = [{"name": f"v{j}", "persistentVolumeClaim": {"claimName": c["metadata"]["name"]}} for j, c in enumerate(claims)]
        by = {o["metadata"]["name"]: o for o in pvs}
        b = types.SimpleNamespace(csiNodeLister=types.SimpleNamespace(Get=lambda name: (None, GoNotFound("csinode.storage.k8s.io", name))),
                                  pvCache=types.SimpleNamespace(GetPV=lambda name: (by[name], None) if name in by else (None, ErrNotFound)),
                                  tryTranslatePVToCSI=lambda logger, pv, csiNode: (pv, None))
        env["CheckNodeAffinity"] = lambda pv, labels: check_node_affinity(env, pv, labels)
        gclaims = [GoObj(Spec=GoObj(VolumeName=c["spec"]["volumeName"])) for c in claims]
        exp = []
        for nd in nodes:
            sat, found, err = env["checkBoundClaims"](b, gclaims, GoObj(Name=nd["name"], Labels=dict(nd["labels"])), None)
            assert err is None
            exp.append(2 if not found else (1 if not sat else 0))  # volume_binding.go:405-418: !boundPVsFound -> ErrReasonPVNotExist; !boundSatisfied -> ErrReasonNodeConflict
        rows["boundClaims"].append({"nodes": nodes, "objs": pvs + claims, "volumes": vols, "verdict": exp})
    return rows


def godiv(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def goint(x):
    return int(x)  # int() truncates toward zero, like Go's conversion (the values here are far inside int64)


def build():
    env = {"math": math, "godiv": godiv, "goint": goint, "MaxNodeScore": PINS["score.max_node_score"], "MaxInt64": (1 << 63) - 1, "MinInt64": -(1 << 63),
           "invalidScore": -1, "parse_int": parse_int, "ErrReasonAffinityRulesNotMatch": PINS["reason.ipa_affinity"],
           "ErrReasonConstraintsNotMatch": PINS["reason.pts_skew"], "ErrReasonNodeLabelNotMatch": PINS["reason.pts_skew"] + PINS["reason.pts_missing_label_suffix"],
           "ErrReasonAntiAffinityRulesNotMatch": PINS["reason.ipa_anti"], "ErrReasonExistingAntiAffinityRulesNotMatch": PINS["reason.ipa_existing_anti"],
           # podMatchesAllAffinityTerms (filtering.go): does the incoming pod match its own required affinity terms -- an input of the harness
           "podMatchesAllAffinityTerms": lambda terms, pod: len(terms) > 0 and pod.self_aff, **{"Resource" + k: v for k, v in RESOURCE_NAMES.items()},
           # (guards only the ignored-resource sets, which Fits passes as nil: fit.go:560-562)
           "IsExtendedResourceName": lambda n: "/" in n, **{"TaintEffect" + k: v for k, v in TAINT_EFFECTS.items()}, **{"SEL_" + k: v for k, v in SELECTION.items()}, "TolerationOpEqual": PINS["toleration.op_equal"], "TolerationOpExists": PINS["toleration.op_exists"],
           "LabelFailureDomainBetaZone": PINS["label.zone_beta"], "LabelTopologyZone": PINS["label.zone"],
           "LabelFailureDomainBetaRegion": PINS["label.region_beta"], "LabelTopologyRegion": PINS["label.region"],
           "minThreshold": PINS["image.min_threshold_mb"] * PINS["image.mb"], "maxContainerThreshold": PINS["image.max_container_threshold_mb"] * PINS["image.mb"],
           "minFeasibleNodesToFind": PINS["search.min_feasible_nodes"], "minFeasibleNodesPercentageToFind": PINS["search.min_feasible_percentage"],
           # round 3: the loop-level pieces
           "GoStruct": GoStruct, "gocopy": gocopy, "GoHeap": GoHeap, "heap": GoContainerHeap, "go_math_log": go_math_log, "MinNodeScore": 0, "NodeInclusionPolicyHonor": "Honor",
           "LabelHostname": "kubernetes.io/hostname", "go_round": go_round, "GoMap": GoMap, "GoPtrMap": GoPtrMap,
           "DefaultBindAllHostIP": "0.0.0.0", "ProtocolTCP": "TCP",
           "newCriticalPaths": lambda: [GoStruct(TopologyValue="", MatchNum=(1 << 31) - 1), GoStruct(TopologyValue="", MatchNum=(1 << 31) - 1)],
           # round 6: the volume plugins
           "GoSet": GoSet, "logger": None, "ErrNotFound": ErrNotFound, "CheckNodeAffinity": None,
           "ErrReasonConflict": "node(s) had no available volume zone", "ErrReasonMaxVolumeCountExceeded": "node(s) exceed max volume count",
           "topologyLabels": [PINS["label.zone_beta"], PINS["label.region_beta"], PINS["label.zone"], PINS["label.region"]]}
    vz = open(os.path.join(REF, S, "framework/plugins/volumezone/volume_zone.go")).read()
    assert "var topologyLabels = []string{\n\tv1.LabelFailureDomainBetaZone,\n\tv1.LabelFailureDomainBetaRegion,\n\tv1.LabelTopologyZone,\n\tv1.LabelTopologyRegion,\n}" in vz
    assert 'ErrReasonConflict = "%s"' % env["ErrReasonConflict"] in vz
    assert 'ErrReasonMaxVolumeCountExceeded = "%s"' % env["ErrReasonMaxVolumeCountExceeded"] in open(os.path.join(REF, S, "framework/plugins/nodevolumelimits/csi.go")).read()
    bsrc = open(os.path.join(REF, S, "framework/plugins/volumebinding/binder.go")).read()
    assert 'ErrReasonNodeConflict ConflictReason = "node(s) didn\'t match PersistentVolume\'s node affinity"' in bsrc and "ErrReasonPVNotExist = \"node(s) unavailable due to one or more pvc(s) bound to non-existent pv(s)\"" in bsrc
    iface = open(os.path.join(REF, S, "framework/interface.go")).read()
    assert re.search(r"MinNodeScore int64 = 0\b", iface) and re.search(r"MaxNodeScore int64 = %d\b" % PINS["score.max_node_score"], iface)
    assert re.search(r'NodeInclusionPolicyHonor NodeInclusionPolicy = "Honor"', open(os.path.join(REF, "vendor/k8s.io/api/core/v1/types.go")).read())
    kt = open(os.path.join(REF, KT)).read()
    assert 'const DefaultBindAllHostIP = "0.0.0.0"' in kt and re.search(r'ProtocolTCP Protocol = "TCP"', open(os.path.join(REF, "vendor/k8s.io/api/core/v1/types.go")).read())
    so = open(os.path.join(REF, S, "schedule_one.go")).read()
    assert "\tprocessedNodes := len(feasibleNodes) + diagnosis.NodeToStatus.Len()\n\tsched.nextStartNodeIndex = (sched.nextStartNodeIndex + processedNodes) % len(allNodes)\n" in so
    assert "\tschedFramework.Parallelizer().Until(ctx, numAllNodes, checkNode, metrics.Filter)\n\tfeasibleNodes = feasibleNodes[:feasibleNodesLen]\n" in so
    assert re.search(r'LabelHostname = "kubernetes.io/hostname"', open(os.path.join(REF, "vendor/k8s.io/api/core/v1/well_known_labels.go")).read())
    ptsf = open(os.path.join(REF, S, "framework/plugins/podtopologyspread/filtering.go")).read()
    assert "return &criticalPaths{{MatchNum: math.MaxInt32}, {MatchNum: math.MaxInt32}}" in ptsf  # newCriticalPaths, as the lambda above has it
    op_src = open(os.path.join(REF, "vendor/k8s.io/apimachinery/pkg/selection/operator.go")).read()
    for k, v in SELECTION.items():
        assert re.search(r"\b%s\s+Operator = \"%s\"" % (k, re.escape(v)), op_src), k
    types_src = open(os.path.join(REF, "vendor/k8s.io/api/core/v1/types.go")).read()
    for k, v in TAINT_EFFECTS.items():
        assert re.search(r"\bTaintEffect%s TaintEffect = \"%s\"" % (k, v), types_src), k
    for k, v in RESOURCE_NAMES.items():
        assert re.search(r"\bResource%s ResourceName = \"%s\"" % (k, v), types_src), k
    sources = {}
    JOINED.clear()
    for entry in FUNCS:
        name, rel, start, params, int_div = entry[:5]
        opts = entry[5] if len(entry) > 5 else {}
        line, body = cut(rel, start.replace("\\t", "\t"), opts.get("nth", 0))
        py = transliterate(name, params, body, int_div, opts)
        exec(py, env)
        if name == "nodeScoreHeap_Less":
            GoHeap.less = staticmethod(env[name])
        sources[name] = {"file": rel, "line": line, "go": "\n".join(body), "python": py}
        if JOINED.get(name):
            sources[name]["joined"] = JOINED[name]
        if opts:
            sources[name]["opts"] = opts
    return env, sources


def vectors(env):
    rnd = random.Random(20260923)
    v = {}
    near = lambda cap, scale: max(0, min(cap, (rnd.randint(0, scale) * cap) // scale + rnd.randint(-2, 2)))
    caps = lambda: rnd.choice([0, 1, 7, 1000, 4000, 15890, 64 << 30, (1 << 40) + 12345, rnd.randint(1, 1 << 45)])
    # leastResourceScorer over 1..4 resources (weights as the config allows them: 1..100)
    rows = []
    for _ in range(1500):
        n = rnd.randint(1, 4)
        al = [caps() for _ in range(n)]
        rq = [rnd.choice([near(a, 100), rnd.randint(0, a + a // 3 + 1), 0, a]) for a in al]
        w = [rnd.randint(1, 100) if rnd.random() < 0.5 else 1 for _ in range(n)]
        rows.append([rq, al, w, env["leastResourceScorer_closure"](rq, al, w)])
    v["leastResourceScorer"] = rows
    rows = []
    for _ in range(2500):
        n = rnd.choice([1, 2, 2, 2, 3, 4, 5])
        al = [caps() for _ in range(n)]
        rq = [rnd.choice([near(a, 200), rnd.randint(0, a + a // 3 + 1), 0, a]) for a in al]
        if n == 2 and rnd.random() < 0.3 and al[0]:
            al[1] = al[0]
            rq[1] = near(al[0], 200)
        rows.append([rq, al, env["balancedResourceScorer"](rq, al)])
    v["balancedResourceScorer"] = rows
    rows = []
    for _ in range(500):
        n = rnd.randint(0, 12)
        top = rnd.choice([0, 1, 3, 100, 8191, 1 << 20])
        sc = [rnd.randint(0, top) for _ in range(n)]
        for reverse in (False, True):
            out = list(sc)
            env["DefaultNormalizeScore"](100, reverse, out)
            rows.append([sc, reverse, out])
    v["DefaultNormalizeScore"] = rows
    rows = []
    for n in list(range(0, 300)) + [rnd.randint(300, 2_000_000) for _ in range(400)] + [5000, 5625, 5750, 6000, 100000, 1000000]:
        for pct in (None, 0, 1, 5, 10, 35, 50, 99, 100):
            sched = types.SimpleNamespace(percentageOfNodesToScore=0)
            rows.append([pct if pct is not None else -1, n, env["numFeasibleNodesToFind"](sched, pct, n)])
    v["numFeasibleNodesToFind"] = rows
    rows = []
    mb = PINS["image.mb"]
    for _ in range(1200):
        k = rnd.randint(0, 4)
        total = rnd.randint(1, 5000)
        sizes = [rnd.choice([rnd.randint(0, 3000 * mb), 23 * mb, 1000 * mb, 40 * mb]) for _ in range(k)]
        nn = [rnd.randint(1, total) for _ in range(k)]
        ncont = rnd.randint(max(1, k), 6)
        s = sum(env["scaledImageScore"](types.SimpleNamespace(NumNodes=a, Size=b), total) for a, b in zip(nn, sizes))
        rows.append([sizes, nn, total, ncont, env["calculatePriority"](s, ncont)])
    v["imageLocality"] = rows
    rows = []
    for _ in range(1200):
        n = rnd.randint(0, 10)
        top = rnd.choice([0, 1, 5, 100, 5000, 1 << 30])
        sc = [rnd.randint(0, top) for _ in range(n)]
        ig = [rnd.random() < 0.2 for _ in range(n)]
        out = list(sc)
        env["ptsNormalizeScore"](out, ig)
        rows.append([sc, [int(x) for x in ig], out])
    v["ptsNormalizeScore"] = rows
    rows = []
    for _ in range(1200):
        n = rnd.randint(1, 10)
        lo = rnd.choice([0, -50, -100000, 7])
        sc = [lo + rnd.randint(0, rnd.choice([0, 1, 3, 100, 99999])) for _ in range(n)]
        out = list(sc)
        env["ipaNormalizeScore"](out)
        rows.append([sc, out])
    v["ipaNormalizeScore"] = rows
    rows = []
    keys, vals_, effs, ops = ["", "dedicated", "gpu"], ["", "infra", "x"], ["", "NoSchedule", "PreferNoSchedule", "NoExecute"], ["", "Equal", "Exists", "Bogus"]
    for _ in range(1500):
        tol = {"Key": rnd.choice(keys), "Value": rnd.choice(vals_), "Effect": rnd.choice(effs), "Operator": rnd.choice(ops)}
        taint = {"Key": rnd.choice(keys[1:]), "Value": rnd.choice(vals_), "Effect": rnd.choice(effs[1:])}
        rows.append([tol, taint, env["ToleratesTaint"](types.SimpleNamespace(**tol), types.SimpleNamespace(**taint))])
    v["ToleratesTaint"] = rows
    rows = []
    lk = [PINS["label.zone_beta"], PINS["label.zone"], PINS["label.region_beta"], PINS["label.region"], "other"]
    for _ in range(600):
        labels = None if rnd.random() < 0.05 else {k: rnd.choice(["", "a", "b:c"]) for k in lk if rnd.random() < 0.5}
        rows.append([labels, env["GetZoneKey"](types.SimpleNamespace(Labels=labels))])
    v["GetZoneKey"] = rows
    rows = []
    label_vals = ["", "a", "b", "3", "10", "-2", "+7", "1_0", "x3", "99999999999999999999"]
    for _ in range(3000):
        op = rnd.choice(list(SELECTION.values()))
        vals = [rnd.choice(label_vals) for _ in range(rnd.choice([0, 1, 1, 2, 3]))]
        ls = {k: rnd.choice(label_vals) for k in ("k", "other") if rnd.random() < 0.7}
        r = types.SimpleNamespace(key="k", operator=op, strValues=vals)
        rows.append([op, vals, ls, env["requirementMatches"](r, ls)])
    v["requirementMatches"] = rows
    rows = []
    keep = env["DoNotScheduleTaintsFilter_closure"]
    for _ in range(1500):
        mk = lambda d: types.SimpleNamespace(**d)
        taints = [{"Key": rnd.choice(keys[1:]), "Value": rnd.choice(vals_), "Effect": rnd.choice(effs[1:])} for _ in range(rnd.choice([0, 1, 2, 3, 5]))]
        tols = [{"Key": rnd.choice(keys), "Value": rnd.choice(vals_), "Effect": rnd.choice(effs), "Operator": rnd.choice(ops[:3])} for _ in range(rnd.choice([0, 0, 1, 2, 4]))]
        tt, tl = [mk(d) for d in taints], [mk(d) for d in tols]
        taint, found = env["FindMatchingUntoleratedTaint"](tt, tl, keep)
        cnt = env["countIntolerableTaintsPreferNoSchedule"](tt, env["getAllTolerationPreferNoSchedule"](tl))
        rows.append([taints, tols, found, tt.index(taint) if found else -1, cnt])
    v["taintVerdict"] = rows
    rows = []
    scal = ["example.com/gpu", "hugepages-2Mi", "vendor.io/fpga"]
    opts = types.SimpleNamespace(EnableDRAExtendedResource=False)
    for _ in range(1500):
        pod = {"cpu": rnd.choice([0, 0, 1, 100, 250, 4000]), "mem": rnd.choice([0, 0, 1, 64 << 20, 1 << 30]), "eph": rnd.choice([0, 0, 0, 1 << 20, 5 << 30]),
               "scalars": {k: rnd.choice([0, 1, 1, 2, 8]) for k in scal if rnd.random() < 0.35}}
        alloc = {"cpu": rnd.choice([0, 100, 4000, 4000, 64000, 64000]), "mem": rnd.choice([0, 64 << 20, 8 << 30, 8 << 30, 64 << 30]), "eph": rnd.choice([0, 1 << 20, 100 << 30, 100 << 30]),
                 "pods": rnd.choice([0, 1, 3, 110, 110]), "scalars": {k: rnd.choice([0, 1, 2, 8, 8]) for k in scal if rnd.random() < 0.7}}
        edge = lambda a, p: max(0, a - max(0, p + rnd.choice([-1, 0, 0, 1, 5, -5]))) if rnd.random() < 0.7 else rnd.randint(0, a) if a else 0
        req = {"cpu": edge(alloc["cpu"], pod["cpu"]), "mem": edge(alloc["mem"], pod["mem"]), "eph": edge(alloc["eph"], pod["eph"]),
               "scalars": {k: edge(v, pod["scalars"].get(k, 0)) for k, v in alloc["scalars"].items() if rnd.random() < 0.8}}
        n_pods = max(0, alloc["pods"] - rnd.choice([0, 1, 1, 2, 50]))
        node = GoNodeInfo(GoResource(alloc["cpu"], alloc["mem"], alloc["eph"], alloc["pods"], alloc["scalars"]), GoResource(req["cpu"], req["mem"], req["eph"], 0, req["scalars"]), n_pods)
        out = env["fitsRequest"](GoPodRequest(pod["cpu"], pod["mem"], pod["eph"], pod["scalars"]), node, GoNilSet(), GoNilSet(), opts)
        rows.append([alloc, req, n_pods, pod, [[r["Reason"], bool(r.get("Unresolvable", False))] for r in out]])
    v["fitsRequest"] = rows
    rows = []
    for _ in range(900):
        # four nodes over two topology keys; per node: existing pods matching ALL the incoming pod's affinity terms, per anti-affinity term the
        # existing pods it matches, per key the existing pods' own anti-affinity terms that match the incoming pod.  The three count maps are put
        # together the way updateWithAffinityTerms / updateWithAntiAffinityTerms do (types.go: one increment per term whose key the node carries)
        labels = [{k: v for k, v in (("zone", rnd.choice(["a", "a", "b", None])), ("host", rnd.choice([f"h{i}", f"h{i}", None]))) if v is not None} for i in range(4)]
        aff_terms = [rnd.choice(["zone", "host"]) for _ in range(rnd.choice([0, 0, 1, 1, 2]))]
        anti_terms = [rnd.choice(["zone", "host"]) for _ in range(rnd.choice([0, 0, 1, 2]))]
        self_aff = rnd.random() < 0.5
        aff_existing = [rnd.choice([0, 0, 0, 1, 2]) if aff_terms else 0 for _ in range(4)]
        anti_existing = [[rnd.choice([0, 0, 0, 1]) for _ in range(4)] for _ in anti_terms]
        exist_anti = {k: [rnd.choice([0, 0, 0, 0, 1]) for _ in range(4)] for k in ("zone", "host") if rnd.random() < 0.4}
        aff, anti, exist = GoMap(), GoMap(), GoMap()
        for i, lb in enumerate(labels):
            for k in aff_terms:
                if k in lb and aff_existing[i]:
                    aff[(k, lb[k])] += aff_existing[i]
            for t, k in enumerate(anti_terms):
                if k in lb and anti_existing[t][i]:
                    anti[(k, lb[k])] += anti_existing[t][i]
            for k, cnt in exist_anti.items():
                if k in lb and cnt[i]:
                    exist[(k, lb[k])] += cnt[i]
        term = lambda k: types.SimpleNamespace(TopologyKey=k)
        pod_info = types.SimpleNamespace(GetRequiredAffinityTerms=lambda a=[term(k) for k in aff_terms]: a, GetRequiredAntiAffinityTerms=lambda a=[term(k) for k in anti_terms]: a,
                                         GetPod=lambda: types.SimpleNamespace(self_aff=self_aff))
        state = types.SimpleNamespace(podInfo=pod_info, affinityCounts=aff, antiAffinityCounts=anti, existingAntiAffinityCounts=exist)
        out = [env["ipaFilter"](state, types.SimpleNamespace(Node=lambda lb=lb: types.SimpleNamespace(Labels=lb))) for lb in labels]
        rows.append([labels, aff_terms, self_aff, aff_existing, anti_terms, anti_existing, exist_anti, out])
    v["ipaFilter"] = rows
    rows = []
    for _ in range(900):
        # four nodes, one or two DoNotSchedule constraints; per node and constraint the existing matching pods.  TpValueToMatchNum[i] sums them per value over
        # the nodes that carry EVERY constraint's key (calPreFilterState, filtering.go:262-296), the critical path's minimum is the smallest of its values
        # (math.MaxInt32 for no domain: newCriticalPaths)
        labels = [{k: v for k, v in (("zone", rnd.choice(["a", "a", "b", "c", None])), ("host", rnd.choice([f"h{i}", f"h{i}", f"h{i}", None]))) if v is not None} for i in range(4)]
        cons = [{"key": k, "maxSkew": rnd.choice([1, 1, 2, 3]), "minDomains": rnd.choice([1, 1, 2, 4, 5]), "selfMatch": rnd.random() < 0.7, "counts": [rnd.choice([0, 0, 1, 2, 3]) for _ in range(4)]}
                for k in rnd.choice([["zone"], ["host"], ["zone", "host"], ["host", "zone"], ["zone", "zone"]])]
        tp = [GoMap() for _ in cons]
        for i, lb in enumerate(labels):
            if all(c["key"] in lb for c in cons):
                for j, c in enumerate(cons):
                    tp[j][lb[c["key"]]] += c["counts"][i]  # (a counted node's domain exists even with 0 matching pods)
        paths = [[types.SimpleNamespace(MatchNum=min(m.values()) if m else (1 << 31) - 1)] for m in tp]
        state = types.SimpleNamespace(Constraints=[types.SimpleNamespace(TopologyKey=c["key"], MaxSkew=c["maxSkew"], MinDomains=c["minDomains"],
                                                                         Selector=types.SimpleNamespace(Matches=lambda _l, m=c["selfMatch"]: m)) for c in cons],
                                      TpValueToMatchNum=tp, CriticalPaths=paths)
        out = [env["ptsFilter"](state, types.SimpleNamespace(Labels=lb), types.SimpleNamespace(Labels={})) for lb in labels]
        rows.append([labels, cons, out])
    v["ptsFilter"] = rows
    # ---- round 3: loop-level pieces ------------------------------------------------------------------------------------------------
    # InterPodAffinity's topology-pair maps: every existing pod of every node goes through updateWithAffinityTerms (all of the incoming pod's
    # affinity terms matched -> one increment per term) and updateWithAntiAffinityTerms (one increment per matched term), as PreFilter applies them
    rows = []
    for _ in range(700):
        n_nodes = rnd.randint(1, 6)
        labels = [{k: v for k, v in (("zone", rnd.choice(["a", "a", "b", "c", None])), ("host", rnd.choice([f"h{i}", f"h{i}", None]))) if v is not None} for i in range(n_nodes)]
        aff_terms = [rnd.choice(["zone", "host"]) for _ in range(rnd.choice([0, 1, 1, 2, 3]))]
        anti_terms = [rnd.choice(["zone", "host"]) for _ in range(rnd.choice([0, 1, 2, 3]))]
        pods = [[[rnd.random() < 0.4, [rnd.random() < 0.35 for _ in anti_terms]] for _ in range(rnd.choice([0, 0, 1, 2, 4]))] for _ in range(n_nodes)]
        aff, anti = GoMap(), GoMap()
        a_terms = [types.SimpleNamespace(TopologyKey=k) for k in aff_terms]
        n_terms = [types.SimpleNamespace(TopologyKey=k, Matches=lambda pod, ns, t=t: pod.anti[t]) for t, k in enumerate(anti_terms)]
        for i, lb in enumerate(labels):
            node = types.SimpleNamespace(Labels=lb)
            for m_all, m_anti in pods[i]:
                existing = types.SimpleNamespace(self_aff=m_all, anti=m_anti)
                env["updateWithAffinityTerms"](aff, a_terms, existing, node, 1)
                env["updateWithAntiAffinityTerms"](anti, n_terms, existing, None, node, 1)
        # and the same pods removed again (value -1: the entries must disappear, not stay at 0)
        aff2, anti2 = GoMap(aff), GoMap(anti)
        for i, lb in enumerate(labels):
            node = types.SimpleNamespace(Labels=lb)
            for m_all, m_anti in pods[i]:
                existing = types.SimpleNamespace(self_aff=m_all, anti=m_anti)
                env["updateWithAffinityTerms"](aff2, a_terms, existing, node, -1)
                env["updateWithAntiAffinityTerms"](anti2, n_terms, existing, None, node, -1)
        assert not aff2 and not anti2
        rows.append([labels, aff_terms, anti_terms, pods, sorted([list(k) + [c] for k, c in aff.items()]), sorted([list(k) + [c] for k, c in anti.items()])])
    v["ipaCountMaps"] = rows
    # PodTopologySpread's calPreFilterState: per-node closure, merge, critical paths -- nodes with / without the keys, node inclusion policies
    # (required node affinity, untolerated NoSchedule taints), pods of other namespaces and terminating pods
    rows = []
    keep = env["DoNotScheduleTaintsFilter_closure"]
    for _ in range(700):
        n_nodes = rnd.randint(1, 7)
        gate = rnd.random() < 0.7  # enableNodeInclusionPolicyInPodTopologySpread
        keys = rnd.choice([["zone"], ["host"], ["zone", "host"], ["zone", "zone"]])
        cons = [{"key": k, "affinityPolicy": rnd.choice(["Honor", "Honor", "Ignore"]), "taintsPolicy": rnd.choice(["Honor", "Ignore", "Ignore"]), "emptySelector": rnd.random() < 0.1} for k in keys]
        nodes = []
        for i in range(n_nodes):
            lb = {k: v for k, v in (("zone", rnd.choice(["a", "a", "b", "c", None])), ("host", rnd.choice([f"h{i}", f"h{i}", f"h{i}", None]))) if v is not None}
            taints = [{"Key": "dedicated", "Value": "infra", "Effect": rnd.choice(["NoSchedule", "PreferNoSchedule", "NoExecute"])}] if rnd.random() < 0.25 else []
            pods = [{"ns": rnd.choice(["default", "default", "other"]), "terminating": rnd.random() < 0.15, "match": [rnd.random() < 0.6 for _ in cons]} for _ in range(rnd.choice([0, 1, 2, 3, 5]))]
            nodes.append({"labels": lb, "taints": taints, "affinityMatch": rnd.random() < 0.8, "pods": pods})
        tolerations = [{"Key": "dedicated", "Value": "infra", "Effect": "", "Operator": "Equal"}] if rnd.random() < 0.3 else []
        mk = lambda d: types.SimpleNamespace(**d)
        pod = types.SimpleNamespace(Namespace="default", Spec=types.SimpleNamespace(Tolerations=[mk(t) for t in tolerations]))
        constraints = [GoStruct(TopologyKey=c["key"], NodeAffinityPolicy=c["affinityPolicy"], NodeTaintsPolicy=c["taintsPolicy"],
                                Selector=types.SimpleNamespace(Empty=lambda e=c["emptySelector"]: e, Matches=lambda lbls, j=j: lbls["match"][j])) for j, c in enumerate(cons)]
        all_nodes = []
        for nd_ in nodes:
            node = types.SimpleNamespace(Labels=nd_["labels"], Spec=types.SimpleNamespace(Taints=[mk(t) for t in nd_["taints"]]), affinityMatch=nd_["affinityMatch"])
            infos = [types.SimpleNamespace(GetPod=lambda p=p: types.SimpleNamespace(DeletionTimestamp=("t" if p["terminating"] else None), Namespace=p["ns"], Labels={"match": p["match"]})) for p in nd_["pods"]]
            all_nodes.append(types.SimpleNamespace(Node=lambda node=node: node, GetPods=lambda infos=infos: infos))
        require = types.SimpleNamespace(Match=lambda node: (node.affinityMatch, None))
        pl = types.SimpleNamespace(enableNodeInclusionPolicyInPodTopologySpread=gate)
        by_node = [[] for _ in all_nodes]
        for n in range(len(all_nodes)):  # parallelizer.Until(ctx, len(allNodes), processNode, ...): the pieces write disjoint slots
            env["calPreFilterState_processNode"](n, pl, pod, all_nodes, constraints, require, by_node)
        st = GoStruct(TpValueToMatchNum=[GoMap() for _ in constraints], CriticalPaths=[None] * len(constraints))
        env["calPreFilterState_merge"](st, by_node)
        env["calPreFilterState_minima"](st, constraints)
        rows.append([gate, cons, nodes, tolerations, [[list(kv) for kv in sorted(m.items())] for m in st.TpValueToMatchNum], [p[0].MatchNum for p in st.CriticalPaths]])
    v["calPreFilterState"] = rows
    # PodTopologySpread's PreScore + Score + NormalizeScore over a FILTERED node list (scoring.go:61-265): ignored nodes, candidate domains and
    # their number -> the log(size + 2) weights, matching pods of ALL nodes counted into the candidate domains (node inclusion policies, other
    # namespaces, terminating pods), hostname constraints scored per node, math.Round, the normalization.  requireAllTopologies = true: the pod
    # carries its own constraints; = false (the plugin's system defaults, scoring.go:140): the second set, at the end of this function
    HOST = env["LabelHostname"]

    def pts_prescore_rows(rnd, require_all, count):
      rows = []
      for _ in range(count):
          n_nodes = rnd.randint(1, 9)
          gate = rnd.random() < 0.7
          keys = rnd.choice([["zone"], [HOST], ["zone", HOST], ["zone", "rack"], ["rack", "zone", HOST], ["zone", "zone"]])
          cons = [{"key": k, "maxSkew": rnd.randint(1, 4), "affinityPolicy": rnd.choice(["Honor", "Honor", "Ignore"]), "taintsPolicy": rnd.choice(["Honor", "Ignore", "Ignore"]),
                   "emptySelector": rnd.random() < 0.1} for k in keys]
          nodes = []
          for i in range(n_nodes):
              lb = {k: v for k, v in (("zone", rnd.choice(["a", "a", "b", "c", None])), ("rack", rnd.choice(["r1", "r2", "r2", None])), (HOST, rnd.choice([f"h{i}"] * 5 + [None]))) if v is not None}
              taints = [{"Key": "dedicated", "Value": "infra", "Effect": rnd.choice(["NoSchedule", "PreferNoSchedule", "NoExecute"])}] if rnd.random() < 0.25 else []
              pods = [{"ns": rnd.choice(["default", "default", "other"]), "terminating": rnd.random() < 0.15, "match": [rnd.random() < 0.6 for _ in cons]} for _ in range(rnd.choice([0, 1, 2, 3, 6]))]
              nodes.append({"name": f"n{i}", "labels": lb, "taints": taints, "affinityMatch": rnd.random() < 0.8, "pods": pods})
          filtered = sorted(rnd.sample(range(n_nodes), rnd.randint(1, n_nodes)))
          tolerations = [{"Key": "dedicated", "Value": "infra", "Effect": "", "Operator": "Equal"}] if rnd.random() < 0.3 else []
          mk = lambda d: types.SimpleNamespace(**d)
          pod = types.SimpleNamespace(Namespace="default", Spec=types.SimpleNamespace(Tolerations=[mk(t) for t in tolerations]))
          constraints = [GoStruct(TopologyKey=c["key"], MaxSkew=c["maxSkew"], NodeAffinityPolicy=c["affinityPolicy"], NodeTaintsPolicy=c["taintsPolicy"],
                                  Selector=types.SimpleNamespace(Empty=lambda e=c["emptySelector"]: e, Matches=lambda lbls, j=j: lbls["match"][j])) for j, c in enumerate(cons)]
          all_nodes = []
          for nd_ in nodes:
              node = types.SimpleNamespace(Name=nd_["name"], Labels=GoLabels(nd_["labels"]), Spec=types.SimpleNamespace(Taints=[mk(t) for t in nd_["taints"]]), affinityMatch=nd_["affinityMatch"])
              infos = [types.SimpleNamespace(GetPod=lambda p=p: types.SimpleNamespace(DeletionTimestamp=("t" if p["terminating"] else None), Namespace=p["ns"], Labels={"match": p["match"]})) for p in nd_["pods"]]
              all_nodes.append(types.SimpleNamespace(Node=lambda node=node: node, GetPods=lambda infos=infos: infos))
          require = types.SimpleNamespace(Match=lambda node: (node.affinityMatch, None))
          pl = types.SimpleNamespace(enableNodeInclusionPolicyInPodTopologySpread=gate)
          st = GoStruct(Constraints=constraints, IgnoredNodes=GoSet(), TopologyValueToPodCounts=[GoPtrMap() for _ in constraints], TopologyNormalizingWeight=[0.0] * len(constraints))
          filtered_infos = [all_nodes[i] for i in filtered]
          topo_size = [0] * len(constraints)
          env["ptsPreScore_initNodes"](st, filtered_infos, require_all, topo_size)
          env["ptsPreScore_weights"](st, filtered_infos, topo_size)
          for n in range(len(all_nodes)):  # parallelizer.Until(ctx, len(allNodes), processAllNode, ...): atomic adds, any order
              env["ptsPreScore_processAllNode"](n, pl, pod, all_nodes, st, require_all, require)
          raw = [env["ptsScore"](st, info.Node(), info, pod)[0] for info in filtered_infos]
          ignored = [info.Node().Name in st.IgnoredNodes for info in filtered_infos]
          norm = list(raw)
          env["ptsNormalizeScore"](norm, ignored)
          rows.append([gate, cons, nodes, tolerations, filtered, [int(x) for x in ignored], [w.hex() for w in st.TopologyNormalizingWeight], raw, norm])
      return rows

    v["ptsPreScoreScore"] = pts_prescore_rows(rnd, True, 700)
    # InterPodAffinity's PreScore + Score + NormalizeScore (interpodaffinity/scoring.go:51-290): the incoming pod's preferred terms against every
    # existing pod, the existing pods' required (HardPodAffinityWeight) and preferred terms against the incoming pod, nodes without labels or
    # without the term's key, only pods with affinity when the incoming pod has no preferred terms, PreScore's Skip when nothing hit
    rows = []
    for _ in range(700):
        n_nodes = rnd.randint(1, 8)
        hard_w = rnd.choice([0, 1, 1, 7])
        mkterm = lambda: {"key": rnd.choice(["zone", "host"]), "weight": rnd.choice([1, 5, 50, 100])}
        inc_aff = [mkterm() for _ in range(rnd.choice([0, 0, 1, 2]))]
        inc_anti = [mkterm() for _ in range(rnd.choice([0, 0, 1, 2]))]
        nodes = []
        for i in range(n_nodes):
            lb = {k: v for k, v in (("zone", rnd.choice(["a", "a", "b", None])), ("host", rnd.choice([f"h{i}"] * 4 + [None]))) if v is not None}
            pods = []
            for _p in range(rnd.choice([0, 1, 2, 4])):
                own = rnd.random() < 0.5  # the pod has affinity terms of its own
                pods.append({"matchAff": [rnd.random() < 0.5 for _ in inc_aff], "matchAnti": [rnd.random() < 0.5 for _ in inc_anti],
                             "required": [dict(mkterm(), matches=rnd.random() < 0.6) for _ in range(rnd.choice([0, 1, 2]) if own else 0)],
                             "prefAff": [dict(mkterm(), matches=rnd.random() < 0.6) for _ in range(rnd.choice([0, 1]) if own else 0)],
                             "prefAnti": [dict(mkterm(), matches=rnd.random() < 0.6) for _ in range(rnd.choice([0, 1]) if own else 0)]})
            nodes.append({"name": f"n{i}", "labels": lb, "pods": pods})
        filtered = sorted(rnd.sample(range(n_nodes), rnd.randint(1, n_nodes)))
        incoming = types.SimpleNamespace(Namespace="default")
        term_on_existing = lambda t, which, j: GoStruct(TopologyKey=t["key"], Matches=lambda pod, ns, which=which, j=j: pod.flags[which][j])
        term_on_incoming = lambda t: GoStruct(TopologyKey=t["key"], Matches=lambda pod, ns, m=t["matches"]: m)
        weighted = lambda term, w: GoStruct(AffinityTerm=term, Weight=w)
        pod_info = types.SimpleNamespace(GetPreferredAffinityTerms=lambda: [weighted(term_on_existing(t, "matchAff", j), t["weight"]) for j, t in enumerate(inc_aff)],
                                         GetPreferredAntiAffinityTerms=lambda: [weighted(term_on_existing(t, "matchAnti", j), t["weight"]) for j, t in enumerate(inc_anti)])
        state = GoStruct(topologyScore=GoPtrMap(), podInfo=pod_info, namespaceLabels=None)
        has_constraints = bool(inc_aff or inc_anti)
        infos = []
        for nd_ in nodes:
            node = types.SimpleNamespace(Name=nd_["name"], Labels=GoLabels(nd_["labels"]))
            pis = []
            for p in nd_["pods"]:
                ep = types.SimpleNamespace(flags=p)
                pis.append(types.SimpleNamespace(GetPod=lambda ep=ep: ep, has_affinity=bool(p["required"] or p["prefAff"] or p["prefAnti"]),
                                                 GetRequiredAffinityTerms=lambda p=p: [term_on_incoming(t) for t in p["required"]],
                                                 GetPreferredAffinityTerms=lambda p=p: [weighted(term_on_incoming(t), t["weight"]) for t in p["prefAff"]],
                                                 GetPreferredAntiAffinityTerms=lambda p=p: [weighted(term_on_incoming(t), t["weight"]) for t in p["prefAnti"]]))
            infos.append(types.SimpleNamespace(Node=lambda node=node: node, GetPods=lambda pis=pis: pis, GetPodsWithAffinity=lambda pis=pis: [x for x in pis if x.has_affinity]))
        # scoring.go:153-165: every node when the incoming pod has preferred terms, else the nodes hosting pods with affinity
        all_nodes = infos if has_constraints else [x for x in infos if x.GetPodsWithAffinity()]
        pl = types.SimpleNamespace(args=types.SimpleNamespace(HardPodAffinityWeight=hard_w))
        topo_scores = []
        for i in range(len(all_nodes)):  # parallelizer.Until(pCtx, len(allNodes), processNode, ...): the maps are merged afterwards, any order
            env["ipaPreScore_processNode"](i, pl, all_nodes, has_constraints, state, incoming, topo_scores)
        skipped = not topo_scores  # index == -1: fwk.Skip
        for ts in topo_scores:     # for i := 0; i <= int(index); i++ { state.topologyScore.append(topoScores[i]) }
            env["scoreMap_append"](state.topologyScore, ts)
        raw = [0 if skipped else env["ipaScore"](state, infos[i].Node())[0] for i in filtered]
        norm = list(raw)
        if not skipped:
            env["ipaNormalizeScore"](norm)
        rows.append([hard_w, inc_aff, inc_anti, nodes, filtered, int(skipped), sorted([k, sorted([list(kv) for kv in m.items()])] for k, m in state.topologyScore.items()), raw, norm])
    v["ipaPreScoreScore"] = rows
    # the node search of one cycle (schedule_one.go:610-693 + :538-539): numNodesToFind, the visiting order from nextStartNodeIndex, the search
    # cancelled by the (K+1)-th feasible node, the nodes processed, the next start index.  ONE worker taking the positions in order (the oracle's
    # canonical mode): parallelizer.Until with one worker checks the context before every piece
    rows = []
    for _ in range(1500):
        n = rnd.choice([rnd.randint(1, 99), rnd.randint(100, 400), rnd.randint(100, 400), 100, 101, 125])
        pct = rnd.choice([0, 0, 5, 30, 50, 99, 100])
        scoring = rnd.random() < 0.85  # a profile without Score plugins keeps ONE feasible node (:619-621)
        start = rnd.randrange(n)
        p_feas = rnd.choice([0.0, 0.02, 0.3, 0.7, 1.0])
        feas = [rnd.random() < p_feas for _ in range(n)]
        num_to_find = env["numFeasibleNodesToFind"](types.SimpleNamespace(percentageOfNodesToScore=0), pct, n)
        if not scoring:
            num_to_find = 1
        ok, bad = types.SimpleNamespace(Code=lambda: "Success", IsSuccess=lambda: True), types.SimpleNamespace(Code=lambda: "Unschedulable", IsSuccess=lambda: False)
        infos = [types.SimpleNamespace(idx=i, Node=lambda i=i: types.SimpleNamespace(Name=f"n{i}")) for i in range(n)]
        fw = types.SimpleNamespace(RunFilterPluginsWithNominatedPods=lambda ctx, state, pod, info: ok if feas[info.idx] else bad)
        sched = types.SimpleNamespace(nextStartNodeIndex=start)
        counter = types.SimpleNamespace(v=0)
        counter.add = lambda d, c=counter: (setattr(c, "v", c.v + d), c.v)[1]
        ctx = types.SimpleNamespace(cancelled=None)
        cancel = lambda why, ctx=ctx: setattr(ctx, "cancelled", why)
        feasible_nodes, result = [None] * num_to_find, [None] * n
        for i in range(n):  # Until(ctx, numAllNodes, checkNode, ...), one worker
            if ctx.cancelled:
                break
            env["findNodesThatPassFilters_checkNode"](i, sched, infos, n, fw, ctx, None, None, None, cancel, counter, num_to_find, feasible_nodes, result)
        feasible_nodes = feasible_nodes[: counter.v]                      # feasibleNodes = feasibleNodes[:feasibleNodesLen]
        processed = len(feasible_nodes) + sum(r is not None for r in result)  # processedNodes := len(feasibleNodes) + diagnosis.NodeToStatus.Len()
        next_start = (start + processed) % n                               # sched.nextStartNodeIndex = (... + processedNodes) % len(allNodes)
        rows.append([n, pct, int(scoring), start, [int(f) for f in feas], [x.idx for x in feasible_nodes], processed, next_start])
    v["findNodesThatPassFilters"] = rows
    # NodePorts: the wanted host ports of a pod against the ports in use on a node -- empty ip / protocol (0.0.0.0 / TCP), the wildcard ip on either
    # side, ports <= 0 (no host port), the same port under another protocol or another ip
    rows = []
    ips, protos, ports = ["", "0.0.0.0", "10.0.0.1", "10.0.0.2"], ["", "TCP", "UDP"], [0, 80, 80, 443, 8080]
    for _ in range(2500):
        mkp = lambda: {"hostIP": rnd.choice(ips), "protocol": rnd.choice(protos), "hostPort": rnd.choice(ports)}
        used, want = [mkp() for _ in range(rnd.choice([0, 1, 2, 4]))], [mkp() for _ in range(rnd.choice([0, 1, 2, 3]))]
        h = {}
        for u in used:  # HostPortInfo.Add (types.go:458-474): port > 0, sanitize, insert
            if u["hostPort"] <= 0:
                continue
            ipp, protop = [u["hostIP"]], [u["protocol"]]
            env["HostPortInfo_sanitize"](h, ipp, protop)
            pp = env["NewProtocolPort"](protop[0], u["hostPort"])
            h.setdefault(ipp[0], {})[(pp.Protocol, pp.Port)] = True
        info = types.SimpleNamespace(GetUsedPorts=lambda h=h: h)
        rows.append([used, want, env["fitsPorts"]([types.SimpleNamespace(HostIP=w["hostIP"], Protocol=w["protocol"], HostPort=w["hostPort"]) for w in want], info)])
    v["fitsPorts"] = rows
    # RunScorePlugins: weight x normalized score per plugin, summed per node
    rows = []
    for _ in range(600):
        n_nodes, n_pl = rnd.randint(1, 8), rnd.randint(1, 7)
        weights = [rnd.choice([1, 1, 2, 3, 10000, 0]) for _ in range(n_pl)]
        scores = [[rnd.randint(0, 100) for _ in range(n_nodes)] for _ in range(n_pl)]
        plugins = [types.SimpleNamespace(Name=lambda j=j: f"p{j}") for j in range(n_pl)]
        f = types.SimpleNamespace(scorePluginWeight={f"p{j}": w for j, w in enumerate(weights)})
        nodes_ = [types.SimpleNamespace(Node=lambda i=i: types.SimpleNamespace(Name=f"n{i}")) for i in range(n_nodes)]
        out, err = [None] * n_nodes, []
        for i in range(n_nodes):
            env["RunScorePlugins_weigh"](i, f, nodes_, plugins, {f"p{j}": scores[j] for j in range(n_pl)}, out, err, None)
        assert not err
        rows.append([weights, scores, [o.TotalScore for o in out]])
    v["RunScorePlugins_weigh"] = rows
    # selectHost: which node can it return?  The reservoir sampling is driven through every outcome (Intn scripted to pick the t-th node of the
    # maximum in heap order); the canonical choice of this engine is the lowest list position among the possible winners
    rows = []
    for _ in range(500):
        n_nodes = rnd.randint(2, 12)
        top = rnd.choice([1, 3, 100, 700])
        totals = [rnd.randint(0, top) for _ in range(n_nodes)]
        possible = set()
        for t in range(0, n_nodes + 1):
            lst = [GoStruct(Name=f"n{i}", TotalScore=sc) for i, sc in enumerate(totals)]
            env["rand"] = ScriptedRand(t)
            name, _lst, err = env["selectHost"](lst, rnd.choice([1, 3, n_nodes]))
            assert err is None
            possible.add(int(name[1:]))
        assert possible == {i for i, sc in enumerate(totals) if sc == max(totals)}  # (every node of the maximum, nothing else)
        rows.append([totals, sorted(possible), min(possible)])
    v["selectHost"] = rows
    v["topologyNormalizingWeight"] = [[n, env["topologyNormalizingWeight"](n).hex()] for n in list(range(0, 4097)) + [10 ** 6, 2 ** 31 - 3]]
    names = ["busybox", "busybox:1.36", "localhost:5000/app", "localhost:5000/app:v2", "gcr.io/x/y@sha256:abc", "a/b/c", "a:b/c", "", ":", "/", "x:", "reg.io:443/ns/img:tag"]
    v["normalizedImageName"] = [[n, env["normalizedImageName"](n)] for n in names]
    # requireAllTopologies = false (scoring.go:140: a pod without constraints of its own under the plugin's system defaults): no node is ignored,
    # a missing key is the value "" when the domains are sized and counted, and scores nothing (a stream of its own: the sets above keep theirs)
    v["ptsPreScoreScoreRelaxed"] = pts_prescore_rows(random.Random(20260924), False, 700)
    for fam, rows in volume_filter_rows(env, random.Random(20260930)).items():  # round 6: the volume plugins' Filters on object graphs
        v["volumeFilters_" + fam] = rows
    return v


if __name__ == "__main__":
    env, sources = build()
    out = {"sources": sources, "vectors": vectors(env)}
    path = os.path.join(HERE, "reference_vectors.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print({k: len(r) for k, r in out["vectors"].items()}, "->", path, os.path.getsize(path) // 1024, "KB")
    for name, s in sources.items():
        print(f"--- {name} ({s['file']}:{s['line']})\n{s['python']}")
