"""Pin the CPU oracle against every known answer the reference holds for this path
(SURVEY 8(c) KA1-KA5).  The reference has no numeric golden files; these are its prose results
(README.md) and the FailType assertions of pkg/framework/simulator_test.go."""
import numpy as np
import helpers as H
from cluster_capacity_amd import model as M, report as R

DEFAULT = M.Profile.default()


def test_ka1_test_prediction_unlimited(ccref):
    # simulator_test.go:162-174 case limit=0 -> FailType "Unschedulable"
    nodes, pod = H.test_prediction_nodes(), H.test_prediction_pod()
    r = ccref.run(DEFAULT, nodes, pod, max_limit=0)
    assert r.placed == 9 and r.per_node_count.tolist() == [3, 3, 3]
    assert r.stop == M.STOP_UNSCHEDULABLE
    msg = R.stop_reason(r, nodes.n, 0)
    assert R.main_fail_reason(msg)["failType"] == "Unschedulable"
    assert msg == ("Unschedulable: 0/3 nodes are available: 1 Insufficient cpu, 3 Too many pods. "
                   "preemption: 0/3 nodes are available: 3 No preemption victims found for incoming pod.")


def test_ka1_test_prediction_limit6(ccref):
    # simulator_test.go case limit=6 -> FailType "LimitReached"
    nodes, pod = H.test_prediction_nodes(), H.test_prediction_pod()
    r = ccref.run(DEFAULT, nodes, pod, max_limit=6)
    assert r.placed == 6 and r.stop == M.STOP_LIMIT
    msg = R.stop_reason(r, nodes.n, 6)
    assert msg == "LimitReached: Maximum number of pods simulated: 6"
    assert R.main_fail_reason(msg) == {"failType": "LimitReached", "failMessage": "Maximum number of pods simulated: 6"}


def test_e2e_limit5(ccref):
    # test/e2e/e2e_test.go:37-38,171: limit 5 -> LimitReached
    r = ccref.run(DEFAULT, H.readme_nodes(2), H.examples_pod(), max_limit=5)
    assert r.placed == 5 and r.stop == M.STOP_LIMIT


def test_ka2_readme_demo_4_nodes(ccref):
    # README.md:44-66: 52 = 13+13+13+13, "Insufficient cpu" on all 4
    nodes = H.readme_nodes(4)
    r = ccref.run(DEFAULT, nodes, H.examples_pod())
    assert r.placed == 52 and r.per_node_count.tolist() == [13] * 4
    assert R.stop_reason(r, 4, 0).startswith("Unschedulable: 0/4 nodes are available: 4 Insufficient cpu.")


def test_ka2_readme_job_2_nodes(ccref):
    # README.md:216-229: 52 = 26+26 needs 4-CPU nodes (26*150m = 3900m)
    nodes = H.simple_nodes([4000, 4000], [int(8e9)] * 2, [110, 110])
    r = ccref.run(DEFAULT, nodes, H.examples_pod())
    assert r.placed == 52 and r.per_node_count.tolist() == [26, 26]
    assert "2 Insufficient cpu" in R.stop_reason(r, 2, 0)


def test_ka5_score_unit_vectors(ccref):
    GiB, MiB = H.GiB, H.MiB
    # LeastAllocated(alloc 4000m/8GiB, nz-req 1000m/2GiB, pod 150m/100Mi) = (71+73)/2 = 72
    assert ccref.least_allocated([1000 + 150, 2 * GiB + 100 * MiB], [4000, 8 * GiB], [1, 1]) == 72
    # Balanced: f=(0.2875, 0.26220703125), std~0.0126 -> int64(98.73..)=98
    assert ccref.balanced_allocation([1150, 2 * GiB + 100 * MiB], [4000, 8 * GiB]) == 98
    # requested > capacity clamps: least -> 0, fraction -> 1
    assert ccref.least_allocated([5000, 0], [4000, 8 * GiB], [1, 1]) == 50
    assert ccref.balanced_allocation([5000, 8 * GiB], [4000, 8 * GiB]) == 100
    # three resources -> population std-dev path
    assert ccref.balanced_allocation([10, 50, 90], [100, 100, 100]) == int((1 - np.sqrt(((0.4) ** 2 * 2) / 3)) * 100)
    # DefaultNormalizeScore
    assert ccref.default_normalize(100, True, [0, 1, 2]) == [100, 50, 0]
    assert ccref.default_normalize(100, False, [0, 10, 40]) == [0, 25, 100]
    assert ccref.default_normalize(100, True, [0, 0]) == [100, 100]
    assert ccref.default_normalize(100, False, [0, 0]) == [0, 0]


def test_num_feasible_nodes_to_find(ccref):
    # schedule_one.go:697-723
    f = ccref.num_feasible_nodes_to_find
    assert f(0, 4) == 4 and f(0, 99) == 99
    assert f(0, 1000) == 420 and f(0, 10_000) == 500 and f(0, 100_000) == 5000 and f(0, 1_000_000) == 50_000
    assert f(100, 1000) == 1000 and f(10, 500) == 100


def test_go_log_matches_libm_closely(ccref):
    import math
    for x in [2.0, 3.0, 5.0, 18.0, 66.0, 1002.0, 1e6 + 2]:
        assert abs(ccref.go_log(x) - math.log(x)) <= 2 * np.spacing(math.log(x))
    assert ccref.go_log(1.0) == 0.0


def test_sampling_mode_b_same_final_distribution(ccref):
    # final count/distribution at exhaustion are identical in mode A (100 %) and
    # mode B (adaptive sampling) for order-independent plugin sets.
    rng = np.random.default_rng(7)
    n = 300
    nodes = H.simple_nodes(rng.choice([2000, 4000, 8000], n), rng.choice([4, 8, 16], n) * H.GiB, [20] * n,
                           req_mcpu=rng.integers(0, 1500, n), req_mem=rng.integers(0, 2, n) * H.GiB)
    pod = H.simple_pod(500, H.GiB)
    a = ccref.run(M.Profile(percentage_of_nodes_to_score=100), nodes, pod)
    b = ccref.run(M.Profile(percentage_of_nodes_to_score=0), nodes, pod)
    assert a.placed == b.placed and np.array_equal(a.per_node_count, b.per_node_count)
    assert b.evaluated_total < a.evaluated_total  # sampling really visits fewer nodes
    assert not np.array_equal(a.log, b.log)       # ... and is order-dependent round by round


def test_scalar_resource_known_answer(ccref):
    # fit.go:617-657: extended resources bind like the native ones.  2 nodes with 2 and 5 "example.com/gpu", pod asks 2:
    # 1 + 2 instances; cpu/memory/pods never bind.  Reason uses the resource's name (fit.go:640-647).
    z = np.zeros(2, np.int64)
    nodes = M.NodesSoA(alloc=[np.array([64000, 64000]), np.array([1 << 40, 1 << 40]), z, np.array([2, 5])], alloc_pods=np.array([110, 110]),
                       req=[z, z, z, z], nz_mcpu=z, nz_mem=z, pod_count=np.zeros(2, np.int32), taintset_id=np.zeros(2, np.int32),
                       unschedulable=np.zeros(2, np.uint8), scalar_names=["example.com/gpu"])
    pod = M.PodSpec(req=np.array([100, 1 << 20, 0, 2]), nz_mcpu=100, nz_mem=1 << 20, has_scalar_entries=True)
    r = ccref.run(DEFAULT, nodes, pod)
    assert r.placed == 3 and r.per_node_count.tolist() == [1, 2]
    assert R.stop_reason(r, 2, 0, scalar_names=nodes.scalar_names).startswith(
        "Unschedulable: 0/2 nodes are available: 2 Insufficient example.com/gpu.")
