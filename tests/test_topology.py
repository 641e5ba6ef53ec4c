"""CPU: rank -> cores of the NUMA node its GPU hangs off (hvqm4_amd/topology.py), against fake sysfs trees."""
import os

from hvqm4_amd import topology


def _write(root, rel, text):
    path = os.path.join(root, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def _fake_host(root, gpu_nodes, cpus_per_node=32, kfd=True):
    """two CPU KFD nodes, then one GPU per entry of gpu_nodes (its NUMA node); GPU g sits at 0000:(0x10+g):00.0"""
    nn = max(gpu_nodes) + 1
    for n in range(nn):
        _write(root, f"devices/system/node/node{n}/cpulist", f"{n * cpus_per_node}-{(n + 1) * cpus_per_node - 1}\n")
        if kfd:
            _write(root, f"class/kfd/kfd/topology/nodes/{n}/properties", "cpu_cores_count 32\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g, node in enumerate(gpu_nodes):
        bus = 0x10 + g
        addr = f"0000:{bus:02x}:00.0"
        _write(root, f"bus/pci/devices/{addr}/numa_node", f"{node}\n")
        if kfd:
            _write(root, f"class/kfd/kfd/topology/nodes/{nn + g}/properties",
                   f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        else:
            _write(root, f"devices/pci0000:00/{addr}/vendor", "0x1002\n")
            os.makedirs(os.path.join(root, "class/drm"), exist_ok=True)
            os.symlink(os.path.join(root, f"devices/pci0000:00/{addr}"), os.path.join(root, f"class/drm/card{g}_dev"))
            os.makedirs(os.path.join(root, f"class/drm/card{g}"), exist_ok=True)
            os.symlink(os.path.join(root, f"devices/pci0000:00/{addr}"), os.path.join(root, f"class/drm/card{g}/device"))


def test_cpulist_forms():
    assert topology.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert topology.parse_cpulist("") == []


def test_eight_gpus_on_two_sockets_split_their_own_nodes_cores(tmp_path):
    root = str(tmp_path)
    _fake_host(root, [0, 0, 0, 0, 1, 1, 1, 1], cpus_per_node=32)
    allowed = list(range(64))
    seen = []
    for r in range(8):
        cores, node, how = topology.rank_cores(r, 8, allowed, root, env={})
        assert how == "numa" and node == (0 if r < 4 else 1)
        assert len(cores) == 8 and all((c // 32) == node for c in cores)
        seen += cores
    assert sorted(seen) == allowed                       # disjoint, complete


def test_interleaved_gpu_to_node_order_and_visibility_filter(tmp_path):
    root = str(tmp_path)
    _fake_host(root, [1, 0, 1, 0], cpus_per_node=16)      # GPU 0 on node 1, GPU 1 on node 0, ...
    assert topology.gpu_numa_node(0, root, {}) == 1 and topology.gpu_numa_node(1, root, {}) == 0
    # ROCR_VISIBLE_DEVICES=2,1: device 0 of the process is GPU 2 (node 1), device 1 is GPU 1 (node 0)
    env = {"ROCR_VISIBLE_DEVICES": "2,1"}
    assert topology.gpu_numa_node(0, root, env) == 1 and topology.gpu_numa_node(1, root, env) == 0
    assert topology.gpu_numa_node(2, root, env) is None
    cores, node, how = topology.rank_cores(1, 2, list(range(32)), root, env)
    assert (node, how) == (0, "numa") and cores == list(range(16))     # alone on node 0: all of its cores


def test_affinity_mask_restricts_and_unknown_nodes_fall_back(tmp_path):
    root = str(tmp_path)
    _fake_host(root, [0, 1], cpus_per_node=8)
    # the process may only run on cores 0-3 and 12-15
    cores, node, how = topology.rank_cores(1, 2, [0, 1, 2, 3, 12, 13, 14, 15], root, env={})
    assert (node, how) == (1, "numa") and cores == [12, 13, 14, 15]
    # a rank whose node has no allowed core gets the linear slice instead
    cores, node, how = topology.rank_cores(1, 2, [0, 1, 2, 3], root, env={})
    assert how == "linear" and cores == [2, 3]
    # no topology at all (a container without /sys/class/kfd): linear, as before round 5
    empty = str(tmp_path / "empty"); os.makedirs(empty)
    cores, node, how = topology.rank_cores(0, 2, list(range(8)), empty, env={})
    assert (node, how) == (None, "linear") and cores == [0, 1, 2, 3]


def test_drm_fallback_when_kfd_topology_is_absent(tmp_path):
    root = str(tmp_path)
    _fake_host(root, [1, 0], cpus_per_node=4, kfd=False)
    assert topology.gpu_pci_addresses(root) == ["0000:10:00.0", "0000:11:00.0"]
    assert topology.gpu_numa_node(0, root, {}) == 1


def test_hip_visible_devices_wins_over_cuda_visible_devices(tmp_path):
    """launchers export both with the same list: the HIP runtime reads HIP_VISIBLE_DEVICES when it is set and CUDA_VISIBLE_DEVICES only
    otherwise -- applying one after the other would index the filtered list a second time and lose every device"""
    assert topology.visible_indices(8, {"HIP_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": "4,5,6,7"}) == [4, 5, 6, 7]
    assert topology.visible_indices(8, {"CUDA_VISIBLE_DEVICES": "6,7"}) == [6, 7]
    assert topology.visible_indices(8, {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "1"}) == [1]
    # ROCR filters first (the layer below), then HIP indexes what is left
    assert topology.visible_indices(8, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1,3", "CUDA_VISIBLE_DEVICES": "0"}) == [5, 7]
    root = str(tmp_path)
    _fake_host(root, [0, 0, 0, 0, 1, 1, 1, 1], cpus_per_node=8)
    env = {"HIP_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": "4,5,6,7"}
    cores, node, how = topology.rank_cores(0, 4, list(range(16)), root, env)
    assert (node, how) == (1, "numa") and cores == [8, 9]


def test_linear_fallback_avoids_the_cores_of_numa_placed_ranks(tmp_path):
    """a mixed host: GPUs 0 and 1 have a known node, GPU 2's numa_node says -1 -- its rank takes cores nobody was placed on"""
    root = str(tmp_path)
    _fake_host(root, [0, 1, 0], cpus_per_node=4)
    _write(root, "bus/pci/devices/0000:12:00.0/numa_node", "-1\n")
    _write(root, "devices/system/node/node2/cpulist", "8-11\n")          # cores of a node no GPU reports
    allowed = list(range(12))
    got = [topology.rank_cores(r, 3, allowed, root, env={}) for r in range(3)]
    assert [g[2] for g in got] == ["numa", "numa", "linear"]
    assert got[0][0] == [0, 1, 2, 3] and got[1][0] == [4, 5, 6, 7] and got[2][0] == [8, 9, 10, 11]
