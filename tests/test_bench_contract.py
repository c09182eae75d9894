"""bench.py contract on the GPU box: the single-process line and the torch.distributed.run launch the driver uses for N > 1
(two ranks sharing the one test GPU over gloo: RTFS_BENCH_ONE_GPU=1; on the 8-GPU node the same code runs over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"}


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


DP_KEYS = {"metric", "value", "unit", "dtype", "ms_per_step", "ms_per_step_median", "steps", "warmup", "workload", "roofline", "n_gpus", "global_batch",
           "parallelism", "dist", "grad_allreduce", "syncbn_collectives", "baseline_config"}
AR_KEYS = {"collective", "backend", "buckets_per_step", "bytes_per_step", "ms_per_step_mean_in_step", "ms_per_bucket_median_in_step", "ms_standalone"}


def _check_allreduce(ar):
    assert AR_KEYS <= set(ar), ar
    # every trainable parameter of RTFS-Net (740,210 floats = 2.96 MB) crosses the collective once per step, in ONE bucket (train.py:135-146: DDP defaults)
    assert ar["buckets_per_step"] == 1 and ar["bytes_per_step"] == 4 * 740210, ar
    assert ar["ms_per_step_mean_in_step"] > 0 and ar["ms_standalone"] > 0


def _check_dp_rider(dp, world=2):
    assert dp is not None and dp.get("value"), dp
    assert DP_KEYS <= set(dp), sorted(DP_KEYS - set(dp))
    assert dp["n_gpus"] == world and dp["global_batch"] == 2 * world and "training step" in dp["workload"] and dp["ms_per_step"] > 0
    assert dp["parallelism"].startswith(f"dp{world}: DistributedDataParallel") and "SyncBatchNorm" in dp["parallelism"]
    assert dp["dist"]["world_size"] == world and dp["dist"]["ranks_reporting"] == world
    _check_allreduce(dp["grad_allreduce"])
    # SyncBatchNorm's own collectives (train.py:145): the VP block's 9 + 10 packed all-reduces, the CAF cell's 1 + 1 + 1, the batch-count probe - a fixed,
    # small number per step whatever the world size (it is what the 8-GPU step adds to the 1-GPU step besides the gradient bucket: DESIGN.md section 6)
    sb = dp["syncbn_collectives"]
    assert 15 <= sb["per_step"] <= 30 and sb["per_step"] == int(sb["per_step"]) and sb["bytes_per_step"] < 1 << 20 and sb["ms_per_step_sum"] > 0, sb


def test_single_process_line():
    r = subprocess.run([sys.executable, "bench.py", "--layers", "2", "--batch", "2", "--steps", "2", "--warmup", "1", "--cpu-budget-s", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _last_json(r.stdout)
    assert KEYS <= set(res) and "cpu_baseline" in res
    assert res["n_gpus"] == 1 and res["scaling"] == "weak" and res["dtype"] == "f32" and res["vs_baseline"] is None
    assert res["value"] > 0 and "workload" in res["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(res["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(res["cpu_baseline"])
    assert res["si_sdri_parity"]["abs_diff_db"] <= 0.01  # BASELINE.json metric: "...; SI-SDRi parity" (HIP vs oracle, same utterance)
    # BASELINE config 3 rides along: the training step of the same configuration, measured by a child `--mode train` run
    tr = res["training_step"]
    assert tr is not None and tr["value"] > 0 and tr["ms_per_step"] > 0 and "training step" in tr["workload"]
    assert tr["roofline"]["bound"] == "mfma" and "wgrad" in tr["roofline"]["kernel"]
    sb = res["split_bf16"]
    assert sb is not None and sb["dtype"] == "bf16x3" and sb["value"] > 0
    s6 = res["split_bf16x6"]
    assert s6 is not None and s6["dtype"] == "bf16x6" and s6["value"] > 0 and "fp32-equivalent" in s6["accuracy"]
    assert "loop" in res["cpu_baseline"]["sru"]  # (which SRU loop the oracle ran: the C restatement where gcc exists, else the Python one)
    tb = res["training_step_split_bf16"]
    assert tb is not None and tb["dtype"] == "bf16x3" and tb["value"] > 0 and "training step" in tb["workload"]
    t6 = res["training_step_split_bf16x6"]
    assert t6 is not None and t6["dtype"] == "bf16x6" and t6["value"] > 0 and "training step" in t6["workload"]
    # the other BASELINE.json configurations ride along at their OWN sizes (bench.py INFER_RIDERS)
    c2, c5, c5b, b1 = res["config2"], res["config5_bf16x3"], res["config5_bf16"], res["latency_b1"]
    assert c2 is not None and "RTFS-Net-4" in c2["workload"] and "batch 16" in c2["workload"] and c2["value"] > 0
    assert c5 is not None and "RTFS-Net-12" in c5["workload"] and "4 s" in c5["workload"] and c5["dtype"] == "bf16x3" and c5["roofline"] is not None
    assert c5b is not None and c5b["dtype"] == "bf16" and c5b["value"] > 0
    assert b1 is not None and "batch 1 " in b1["workload"] and 0 < b1["ms_per_utterance"] < 64.7


def test_lip_encoder_in_the_timed_step():
    r = subprocess.run([sys.executable, "bench.py", "--lip", "--layers", "2", "--batch", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _last_json(r.stdout)
    assert KEYS <= set(res) and res["config"]["mode"] == "infer+lip-encoder" and res["value"] > 0


def test_plain_launch_with_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run in front (the form the driver uses for N = 1): the script re-executes itself
    under torch.distributed.run instead of exiting.  (`--no-train-line`: the DDP rider of the N > 1 line is test_two_rank_launch's subject.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RTFS_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--layers", "2", "--batch", "2", "--steps", "2", "--warmup", "1", "--no-train-line"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = _last_json(r.stdout)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["value"] > 0
    assert "no data-path collective" in res["config"]["parallelism"] and "training_step_dp" not in res


def test_two_rank_launch():
    """the driver's N > 1 launch form (two ranks sharing the test GPU over gloo): the headline stays the inference line over utterance shards, and BASELINE
    config 4 rides along as `training_step_dp` - a CHILD torch.distributed.run of `bench.py --mode train` on the same ranks (DDP + SyncBatchNorm step, one
    gradient bucket, all-reduce timed), whose own JSON line this test therefore covers as well"""
    env = dict(os.environ, RTFS_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29531", "bench.py", "--gpus", "2", "--layers", "2", "--batch", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = _last_json(r.stdout)
    assert KEYS <= set(res) and "cpu_baseline" not in res  # the CPU baseline is a rank-0, N = 1 measurement
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["value"] > 0
    assert res["dist"]["world_size"] == 2 and res["dist"]["ranks_reporting"] == 2
    assert "no data-path collective" in res["config"]["parallelism"]
    _check_dp_rider(res["training_step_dp"])
    assert "no scaling curve" in res["scaling_note"]


def test_four_rank_launch_dry_run_of_the_eight_gpu_line():
    """the 8-GPU run is one shot per round, so its code path is exercised here with MORE than two ranks - four ranks share the test GPU
    (RTFS_BENCH_ONE_GPU=1, gloo), rank 0 of the parent still holds its HIP context while the child torch.distributed.run of the DDP step starts four more
    processes on the same GPU - and against the wall-clock budget the line reports (< 600 s for `--gpus 8`; this dry run < 300 s)."""
    import time

    env = dict(os.environ, RTFS_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port",
           "29537", "bench.py", "--gpus", "4", "--layers", "2", "--batch", "2", "--steps", "2", "--warmup", "1"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = _last_json(r.stdout)
    assert res["n_gpus"] == 4 and res["config"]["global_batch"] == 8 and res["dist"]["ranks_reporting"] == 4 and res["value"] > 0
    _check_dp_rider(res["training_step_dp"], world=4)
    wc = res["wall_clock_s"]
    assert wc["budget"] == 600 and wc["child_timeout"] < wc["budget"] and wc["inference_part"] + wc["training_step_dp_child"] <= wall + 1
    assert wall < 300, (wall, wc)
    print(f"4-rank dry run: {wall:.0f} s wall clock ({wc})")
