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
