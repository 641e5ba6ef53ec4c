"""GPU: both workgroup shapes of the reconstruction kernel on every parity clip.

The runtime picks one or two tiles per workgroup per launch from the AOT payload density (hvq_runtime.cpp, flush_end); the
test clips are dense synthetic streams, so the default run exercises mostly the one-tile kernels.  Here the parity, batch and
GPU-parse suites run once more in a child process with each shape forced (HVQM4_AMD_TILES_PER_WG is read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# tests of the device PARSER alone (17 s of crafted prefix trees) and the 4K clip (11 s of clip generation): the reconstruction modes below
# do not touch what the former exercise, the latter runs in the main suite
NOT_PARSER_ONLY = "not pathological_trees and not 4096-2176"


@pytest.mark.parametrize("tiles", ["1", "2"])
def test_parity_suites_with_forced_tiles_per_workgroup(tiles):
    env = dict(os.environ, HVQM4_AMD_TILES_PER_WG=tiles)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py", "tests/test_gpu_batch.py", "tests/test_gpu_gparse.py", "tests/test_gpu_configs.py",
                        "tests/test_gpu_reject.py", "-k", NOT_PARSER_ONLY],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("cap", ["24", "200"])
def test_parity_suites_with_tiles_beyond_the_pair_list(cap):
    """A tile with more (item, basis) pairs than the picture's pair list reserves (1024 at most) gets no pair list: its items walk
    their bases themselves (the kernel's serial pair phase).
    No encoder-like stream reaches 1024 pairs in 256 blocks; with the list capped at 24 nearly every tile of the parity clips
    takes that path, at 200 the two kinds of tile meet inside one workgroup."""
    env = dict(os.environ, HVQM4_AMD_PAIR_CAP=cap)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py", "tests/test_gpu_gparse.py", "-k", NOT_PARSER_ONLY],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("cap", ["0", "40"])
def test_parity_suites_with_tiles_beyond_the_staged_pool(cap):
    """hvq_recon_inline_kernel stages a tile's range of the payload pool in LDS; a tile whose payload exceeds the launch's share
    reads the rest from HBM.  With the share capped at 40 dwords most tiles of the parity clips mix both, at 0 nothing is staged."""
    env = dict(os.environ, HVQM4_AMD_POOL_CAP=cap)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py", "tests/test_gpu_gparse.py", "-k", NOT_PARSER_ONLY],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


def test_parity_suites_with_two_launch_queues_forced():
    """A batch of 16 streams or more deals its streams -- by work -- to two launch queues (hvq_runtime.cpp build_tiles); the parity
    clips are single streams, so the parity, batch and configuration suites run once more with HVQM4_AMD_QUEUES=2: every batch split,
    self-referencing P pictures included (the randomized sweep has a two-queue mode of its own in tools/sweep.sh)."""
    env = dict(os.environ, HVQM4_AMD_QUEUES="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py", "tests/test_gpu_batch.py", "tests/test_gpu_configs.py", "-k", "not sweep"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


def test_gpu_parse_suite_with_the_chains_only():
    """The device entropy parse has two ways to a blob: the flat path (all sections at once, scans) and round 1's chains,
    which the flat path also falls back to per picture.  The default run exercises the flat path; here the GPU-parse suite
    runs with HVQM4_AMD_PARSE_FLAT=0 so that the chains stay covered as a whole."""
    env = dict(os.environ, HVQM4_AMD_PARSE_FLAT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_gparse.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


def test_flat_parse_kernel_and_chains_kernel_agree_on_corrupted_pictures():
    """108 corrupted pictures (bit flips, truncations, spliced garbage) behind intact ones: the flat parse kernel (with its
    hand-over) and the chains kernel must reject the same ones with the same error and decode the others to the same
    pictures."""
    outs = []
    for flat in ("1", "0"):
        env = dict(os.environ, HVQM4_AMD_PARSE_FLAT=flat, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "corrupt_hashes.py")], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        outs.append(r.stdout.strip().splitlines())
    assert len(outs[0]) == 3 * 3 * 12
    assert outs[0] == outs[1], [(a, b) for a, b in zip(outs[0], outs[1]) if a != b][:5]
