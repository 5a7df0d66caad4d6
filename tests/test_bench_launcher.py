"""bench.py's own launcher: `python bench.py --gpus N` without WORLD_SIZE must start N ranks and print ONE JSON line
with n_gpus = N.  Rehearsed here on CPU: backend gloo, decoder replaced by bench.py's byte-copy stand-in (--stub),
so what runs is the real spawn path, rendezvous, barriers, max-over-ranks timing and - in the scatter modes - the
real collectives with a routing check."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--stub", "--steps", "2",
                        "--warmup", "1", "--frames", "48"] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,extra", [("shard", []), ("scatter", ["--chunk-frames", "10"]),
                                        ("scatter", ["--chunk-frames", "8", "--root-frames", "20"]),
                                        ("scatter-plain", [])])
def test_bare_command_spawns_two_ranks(mode, extra):
    r = _run(["--gpus", "2", "--mode", mode] + extra)
    assert r["n_gpus"] == 2 and r["stub"] is True and r["steps"] == 2
    assert r["config"]["launch"] == "spawned by bench.py"
    assert r["scaling"] == "weak" and r["unit"] == "Mbit/s"
    if mode != "shard":
        assert r["routing_ok"] is True


def test_n_gt_1_record_describes_itself():
    """the default N > 1 run (shard mode) also pushes the same frames through the one-root scatter pipeline and prints
    both in ONE JSON line: `value` = per-GPU ingestion, `scatter` = rank 0 owns the stream, with the split it used
    and the bound DESIGN.md (e) predicts; gloo rehearsal, 3 ranks"""
    r = _run(["--gpus", "3", "--frames", "64"])
    assert r["n_gpus"] == 3 and r["mode"] == "shard" and r["backend"] == "gloo" and r["rccl_ranks"] == 0
    assert r["xgmi_bytes_in_timed_region"] == 0
    sc = r["scatter"]
    assert sc["frames"] == 3 * 64 and sc["every_frame_matches_the_shard_decode"] is True and sc["value"] > 0
    assert sc["chunk_frames"] >= 4 and sc["root_frames"] == 2 * sc["chunk_frames"]  # the rehearsal pretends D = 2 L
    assert sc["predicted"]["speedup_vs_1gpu_bound"] == 2.0 and "speedup_vs_1gpu_measured" in sc
    # the leg can be switched off, and the explicit scatter mode does not nest it
    assert "scatter" not in _run(["--gpus", "2", "--no-scatter-leg"])
    assert "scatter" not in _run(["--gpus", "2", "--mode", "scatter", "--chunk-frames", "10"])


def test_hung_scatter_leg_is_a_failure_with_the_shard_line_intact():
    """a collective leg that does not come back: the watchdog prints the shard line WITH the error in it and every rank
    exits non-zero (round-3 advisor: a hang must not end as rc 0)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VIT_BENCH_TEST_HANG_SCATTER"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--stub", "--steps", "2",
                        "--warmup", "1", "--frames", "48", "--gpus", "2", "--scatter-timeout", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and "did not finish" in r["scatter"]["error"]


def test_three_ranks_and_forced_spawn_of_one():
    assert _run(["--gpus", "3", "--mode", "scatter", "--chunk-frames", "7"])["n_gpus"] == 3
    r = _run(["--gpus", "1", "--spawn"])
    assert r["n_gpus"] == 1 and r["config"]["launch"] == "spawned by bench.py"


def test_under_an_external_launcher_it_is_a_rank():
    """WORLD_SIZE in the environment (torch.distributed.run): no second level of children"""
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": "29731"})
    assert r["n_gpus"] == 1 and r["config"]["launch"] == "external launcher"


def test_under_torch_distributed_run_with_two_ranks():
    """the driver's N > 1 command line: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29779", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--stub", "--frames", "32"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["launch"] == "external launcher"


def test_second_stage_input_generator_agrees_with_the_oracle():
    """bench.py builds its RS(120,110) superframes from the code's definition (it may not use the oracle for its
    product-side legs); here the oracle decodes them: valid codewords, one correction per injected error."""
    import numpy as np
    import _vitpkg
    import bench
    O = _vitpkg.load_oracle()
    for nsf, rsdims, p_err in ((24, 24, 0.06), (16, 7, 0.5), (8, 1, 1.0)):
        p, want_out, want_ret = bench.rs_test_block(nsf, rsdims, p_err, seed=nsf + rsdims)
        ret, out = O.rs_check_batch(p, rsdims)
        assert np.array_equal(ret, want_ret) and np.array_equal(out, want_out)
