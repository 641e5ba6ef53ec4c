"""CPU: bench.py's self-started multi-rank mode (`python bench.py --gpus N` with no launcher): the parent starts one
child per GPU with the torchrun environment, forwards rank 0's line and propagates a failing rank's exit code --
without importing torch or touching a GPU itself.  The C4 clip set is checked for its mix per rank."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent(r'''
    import json, os, sys, time
    rank = int(os.environ["RANK"])
    rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HVQM4_DIST_BACKEND")}
    rec["argv"] = sys.argv[1:]
    rec["torch_loaded_in_parent"] = os.environ.get("PARENT_TORCH")
    open(os.path.join(os.environ["STUB_DIR"], f"rank{rank}.json"), "w").write(json.dumps(rec))
    fail = os.environ.get("STUB_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        sys.exit(7)
    if fail is not None:
        time.sleep(30)          # the other ranks would sit at a barrier: the parent must end them
    if rank == 0:
        print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}))
''')

PARENT = textwrap.dedent(r'''
    import os, sys
    sys.path.insert(0, os.environ["HVQ_ROOT"])
    import bench
    rc = bench.launch_ranks(int(sys.argv[1]), ["--steps", "2"], child=[sys.executable, os.environ["STUB"]])
    assert "torch" not in sys.modules and "hvqm4_amd._lib" not in sys.modules, "the parent must not load torch or the HIP library"
    sys.exit(rc)
''')


def _run(tmp_path, n, fail_rank=None, share=False):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ, HVQ_ROOT=ROOT, STUB=str(stub), STUB_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    if fail_rank is not None:
        env["STUB_FAIL_RANK"] = str(fail_rank)
    if share:
        env["HVQM4_BENCH_SHARE_GPU"] = "1"
    return subprocess.run([sys.executable, "-c", PARENT, str(n)], env=env, capture_output=True, text=True, timeout=120)


def test_parent_starts_ranks_with_torchrun_environment_and_forwards_rank0(tmp_path):
    r = _run(tmp_path, 3)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"metric": "stub", "n_gpus": 3}
    ports = set()
    for k in range(3):
        rec = json.loads((tmp_path / f"rank{k}.json").read_text())
        assert rec["RANK"] == str(k) and rec["LOCAL_RANK"] == str(k) and rec["WORLD_SIZE"] == "3"
        assert rec["MASTER_ADDR"] == "127.0.0.1" and rec["argv"] == ["--steps", "2"]
        assert rec["HVQM4_DIST_BACKEND"] is None        # the launcher sets none: Group defaults to gloo
        ports.add(rec["MASTER_PORT"])
    assert len(ports) == 1


def test_ranks_meet_over_gloo_unless_told_otherwise(tmp_path, monkeypatch):
    """the data path has no collective, so the barrier / end-of-run reduce default to gloo (hvqm4_amd/distrib.py): the launcher
    sets no backend, and Group picks gloo whatever the device situation"""
    r = _run(tmp_path, 2, share=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads((tmp_path / "rank1.json").read_text())["HVQM4_DIST_BACKEND"] in (None, "", "gloo")
    src = open(os.path.join(ROOT, "hvqm4_amd", "distrib.py")).read()
    assert 'os.environ.get("HVQM4_DIST_BACKEND") or "gloo"' in src


def test_failing_rank_makes_the_parent_fail_and_ends_the_others(tmp_path):
    r = _run(tmp_path, 3, fail_rank=1)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""                      # no result line from a failed run
    assert "rank 1 exited with code 7" in r.stderr


def test_c4_clip_set_is_balanced_over_i_mod_n_shards():
    sys.path.insert(0, ROOT)
    import bench
    from hvqm4_amd.distrib import shard
    cfgs = [bench.c4_clip_config(i) for i in range(64)]
    assert sum(c.width == 320 for c in cfgs) == 32 and sum(c.width == 640 for c in cfgs) == 32
    assert sum(c.version == "1.3" for c in cfgs) == 32
    assert [c.seed for c in cfgs] == list(range(64)) and all(c.gop == bench.GOP16 and c.repeat_gops == 4 for c in cfgs)
    for n in (1, 2, 4, 8):
        for r in range(n):
            mine = [cfgs[i] for i in shard(64, r, n)]
            combos = {(c.width, c.version) for c in mine}
            assert combos == {(320, "1.3"), (320, "1.5"), (640, "1.3"), (640, "1.5")}, (n, r)
            assert sum(c.width == 320 for c in mine) * 2 == len(mine)


def _fake_eight_gpu_host(root, allowed):
    """a two-socket host with eight GPUs (four per NUMA node) whose cpulists are the cores this process may really run on, so that the
    ranks' sched_setaffinity calls succeed on the test box"""
    from tests.test_topology import _write
    half = len(allowed) // 2
    lists = [allowed[:half], allowed[half:]]
    for n in range(2):
        _write(root, f"devices/system/node/node{n}/cpulist", ",".join(str(c) for c in lists[n]) + "\n")
        _write(root, f"class/kfd/kfd/topology/nodes/{n}/properties", "cpu_cores_count 32\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g in range(8):
        bus = 0x10 + g
        _write(root, f"bus/pci/devices/0000:{bus:02x}:00.0/numa_node", f"{g // 4}\n")
        _write(root, f"class/kfd/kfd/topology/nodes/{2 + g}/properties", f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
    return lists


def _dry(tmp_path, workload):
    allowed = sorted(os.sched_getaffinity(0))
    lists = _fake_eight_gpu_host(str(tmp_path), allowed)
    env = dict(os.environ, HVQM4_AMD_SYSFS=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "HVQM4_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--workload", workload],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), lists


def test_eight_rank_dry_run_of_config_4_from_a_cold_start(tmp_path):
    """`python bench.py --gpus 8 --workload c4 --dry-run`: eight real processes, gloo rendezvous, barriers and reduces, core slices
    from a fake 8-GPU sysfs tree, clip i -> rank i mod 8 -- everything but the GPU work, so the first real SCALE run cannot die in
    plumbing"""
    line, lists = _dry(tmp_path, "c4")
    assert line["dry_run"] and line["n_gpus"] == 8 and line["scaling"] == "strong" and len(line["ranks"]) == 8
    seen = []
    for r, rec in enumerate(line["ranks"]):
        assert rec["rank"] == r and rec["device"] == r
        assert [u["clip"] for u in rec["units"]] == list(range(r, 64, 8))
        assert {(u["w"], u["version"]) for u in rec["units"]} == {(320, "1.3"), (320, "1.5"), (640, "1.3"), (640, "1.5")}
        assert rec["numa_node"] == r // 4
        if len(lists[r // 4]) >= 4:
            assert rec["core_choice"] == "numa" and set(rec["cores"]) <= set(lists[r // 4])
        seen += [u["clip"] for u in rec["units"]]
    assert sorted(seen) == list(range(64))
    assert line["pictures_per_step"] == 64 * 64
    assert line["pixels_per_step"] == 32 * 64 * (320 * 240 + 640 * 480)


def test_eight_rank_dry_run_of_config_5(tmp_path):
    line, _lists = _dry(tmp_path, "c5")
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["pictures_per_step"] == 1024 * 16 and line["pixels_per_step"] == 1024 * 16 * 640 * 480
    seeds = [s for rec in line["ranks"] for s in rec["units"]["clip_seeds"]]
    assert len(set(seeds)) == 64                               # every rank decodes its own clips
    assert [rec["units"]["first_global_stream"] for rec in line["ranks"]] == [128 * r for r in range(8)]
