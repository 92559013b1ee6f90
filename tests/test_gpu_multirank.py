"""The N > 1 bench path on the hardware there is (round-4 verdict, item 7): two ranks under torch.distributed.run, BOTH on GPU 0 — so that
the first SCALE run on a real 8-GPU node cannot die on plumbing: narrow_to_own_gpu, the gloo barrier and MAX, the CPU binding, the
per-rank front-end, and that rank 0 prints the short line of record.  Reference analogue of the path: instances interleaved across
devices, /root/reference/src/qatseqprod.c:601-630; no collective on the data path."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_two_rank_bench_both_ranks_on_gpu0(gpu_plugin):
    env = dict(os.environ, QZ_BENCH_RANK_DEVICES="0,0", QZSTD_HIP_HW_QUEUES="8", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu",
           "--blocks", "256", "--e2e-blocks", "256"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    last = lines[-1]
    assert last.startswith("{") and len(last) < 4096, last[:300]
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["unit"] == "MB/s" and line["dtype"] == "u8"
    cfg = line["config"]
    assert cfg["producer_errors"] == 0 and cfg["roundtrip_sampled"] == "PASS", cfg
    assert cfg["chunks_per_gpu_per_step"] == 256 and cfg["block_bytes"] == 131072 and cfg["level"] == 1
    assert "cpu_binding_rank0" in cfg and "bound" in cfg["cpu_binding_rank0"], cfg
    assert line["roofline"]["kernel_ms_avg"] > 0 and line["roofline"]["bound"] == "hbm"
    assert line["cpu_baseline"] is None  # the CPU legs run at N = 1 only (bench contract)
    assert len(lines) == 1  # stdout carries exactly one line: rank 0's


@pytest.mark.gpu
def test_single_rank_bench_stdout_is_exactly_the_line_of_record(gpu_plugin):
    """N = 1, a small shape: stdout carries ONE line — the short JSON line with the contract's keys — and nothing else (round 4's 20 KB
    line came back from the driver as parsed = null; the side legs live in bench_details.json and on stderr)"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--blocks", "512",
                          "--e2e-blocks", "512"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, out.stdout[:500]
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "vs_cpu_baseline", "details_file"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["config"]["libzstd_fast_build"] is True and line["config"]["producer_errors"] == 0
    assert line["roofline"]["frac"] > 0 and line["roofline"]["peak"] == 8000.0
    assert os.path.isfile(os.path.join(ROOT, line["details_file"]))
