"""Multi-process runs of the denoise paths on real devices, each checked bit for bit against the
single-process loop inside tests/multigpu_worker.py:

* image sharding (ragged: 3 images on 2 ranks) + the single all-gather of final latents;
* the CFG-pair split (one rank runs the conditional, the other the unconditional forward; one all-gather of the two noise
  predictions per step) for Qwen-Image, with and without TeaCache (per-branch decisions equal the single-GPU trace), and
  for SD3.5 (each rank one row of the batch-2 forward).

`*_one_device` variants put both ranks on cuda:0 over gloo, so the protocol is exercised on a 1-GPU box; the `*_rccl`
variants need two GPUs and run over RCCL / xGMI.  The worker is a separate process tree started before anything touches
the GPU there (the reference forks a process per GPU the same way, examples/evaluation/clip_score.py:346-365)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multigpu_worker.py")
MODES = ["flux_shard", "qwen_pair", "qwen_pair_teacache", "sd3_pair"]


def _run(mode, *extra):
    res = subprocess.run([sys.executable, WORKER, "--mode", mode, *extra], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "MULTIGPU_OK" in res.stdout, f"{mode} {extra}:\n{res.stdout[-2000:]}\n{res.stderr[-4000:]}"


@pytest.mark.parametrize("mode", MODES)
def test_two_ranks_one_device_gloo(mode):
    _run(mode, "--world", "2", "--backend", "gloo", "--same-device")


@pytest.mark.parametrize("mode", ["flux_shard", "qwen_pair"])
def test_eight_ranks_one_device_gloo(mode):
    """BASELINE.json's 8-GPU layouts with REAL HIP forwards on one device: dp8 over 9 images (rank 0 owns two), and four CFG
    pairs; every rank's gathered latents equal the single-process loop bit for bit."""
    _run(mode, "--world", "8", "--backend", "gloo", "--same-device")


def test_bench_qwen_four_cfg_pairs_eight_ranks_one_device():
    """`bench.py --model qwen --gpus 8 --cfg-pair` -- the launch line of BASELINE.json configs[3] -- as a plain command on a 1-GPU
    box: 8 spawned ranks on cuda:0 over gloo, reduced depth; one JSON line from rank 0 that says what it is."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "qwen", "--gpus", "8", "--cfg-pair", "--same-device", "--layers", "2",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"], capture_output=True, text=True, timeout=1800, env=env)
    assert res.returncode == 0, res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["config"]["ranks_in_process_group"] == 8 and line["config"]["global_batch"] == 4 and line["value"] > 0
    assert "PROTOCOL TEST" in line["scaling_note"] and "REDUCED DEPTH" in line["config"]["workload"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL)")
@pytest.mark.parametrize("mode", MODES)
def test_two_ranks_rccl(mode):
    _run(mode, "--world", "2", "--backend", "nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 4, reason="needs 4 GPUs (two CFG pairs)")
def test_two_cfg_pairs_rccl():
    _run("qwen_pair", "--world", "4", "--backend", "nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as a plain command: one JSON line from rank 0, n_gpus = ranks seen by RCCL."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=1800)
    assert res.returncode == 0, res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks_in_process_group"] == 2 and line["value"] > 0


@pytest.mark.parametrize("model", ["flux", "sd35", "sdxl"])
def test_bench_collectives_over_rccl_single_rank(model):
    """FDM_FORCE_DIST=1: bench.py initialises a one-rank RCCL process group, so its barrier, max-over-ranks all-reduce and the
    final all-gather of latents run through RCCL on a 1-GPU box too.  sd35 / sdxl carry a batch-2 CFG tensor of which only the
    image's own row is gathered (the gather used to receive both rows and raise)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", model, "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-kernel-timing"], capture_output=True, text=True, timeout=1800, env=env)
    assert res.returncode == 0, res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["ranks_in_process_group"] == 1 and line["value"] > 0
