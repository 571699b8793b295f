"""world_size-2 `gloo` test of the multi-GPU leg on CPU: contiguous channel sharding and the single
gather of the reduced TFR product to rank 0 (quantum_inferno_amd/dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_ch, n_b, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quantum_inferno_amd import dist as qdist
    from quantum_inferno_amd.engine import TfrResult

    first, count = qdist.shard(total_ch, rank, world)
    results = []
    for panel in range(2):  # CWT and STX reduced products of this rank's channels
        ch = torch.arange(first, first + count, dtype=torch.float64).reshape(-1, 1)
        band = ch * 1000 + panel * 100 + torch.arange(n_b, dtype=torch.float64)[None, :]
        time = (ch * 10 + panel + torch.arange(n, dtype=torch.float64)[None, :] / n).to(torch.float32)
        stats = torch.cat([ch + panel, ch * 2, ch * 3, torch.zeros_like(ch)], dim=1)
        results.append(TfrResult(frequency_hz=np.arange(n_b), power_band=band, power_time=time, stats=stats))
    flat = qdist.pack_reduced(results)
    got = qdist.gather_reduced(flat, dst=0)
    if rank == 0:
        torch.save(got, os.path.join(out_dir, "gathered.pt"))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, steps, length, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quantum_inferno_amd import dist as qdist

    depth = 2
    messages = [torch.empty(length, dtype=torch.float64) for _ in range(depth)]
    pipe = qdist.GatherPipeline(depth=depth, dst=0)
    seen = []
    for k in range(steps):
        i = pipe.acquire()  # the gather that last read messages[i] is over: it may be written again
        if rank == 0 and pipe.outs[i] is not None:
            seen.append(pipe.outs[i].clone())  # what step k - depth delivered
        messages[i].copy_(torch.arange(length, dtype=torch.float64) + 1000.0 * rank + 10.0 * k)
        pipe.submit(i, messages[i])
    outs = pipe.drain()
    if rank == 0:
        torch.save({"seen": seen, "last": [o.clone() for o in outs]}, os.path.join(out_dir, "pipe.pt"))
    else:
        assert all(o is None for o in outs)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pipeline_world2(tmp_path):
    """GatherPipeline (the bench's overlapped gather): every step's message of every rank arrives intact although the
    buffers are reused every second step."""
    steps, length, world = 5, 64, 2
    mp.spawn(_pipeline_worker, args=(world, _free_port(), steps, length, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "pipe.pt"))
    base = torch.arange(length, dtype=torch.float64)
    for k, buf in enumerate(got["seen"]):  # steps 0 .. steps - 3
        for rank in range(world):
            assert torch.equal(buf[rank], base + 1000.0 * rank + 10.0 * k)
    for k in (steps - 2, steps - 1):
        for rank in range(world):
            assert torch.equal(got["last"][k % 2][rank], base + 1000.0 * rank + 10.0 * k)


def test_shard_partitions_every_channel_once():
    from quantum_inferno_amd import dist as qdist

    for total in (1, 7, 8, 64, 512, 1024):
        for world in (1, 2, 3, 8):
            spans = [qdist.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_gather_of_reduced_product_world2(tmp_path):
    total_ch, n_b, n, world = 6, 5, 16, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total_ch, n_b, n, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "gathered.pt"))
    sys.path.insert(0, ROOT)
    from quantum_inferno_amd import dist as qdist

    assert got.shape == (world, 2 * qdist.reduced_slots(3, n_b, n, torch.float32))
    for rank in range(world):
        first, count = qdist.shard(total_ch, rank, world)
        parts = qdist.unpack_reduced(got[rank], count, [(n_b, n), (n_b, n)])
        for panel, (band, time, stats) in enumerate(parts):
            ch = torch.arange(first, first + count, dtype=torch.float64).reshape(-1, 1)
            assert torch.equal(band, ch * 1000 + panel * 100 + torch.arange(n_b, dtype=torch.float64)[None, :])
            assert time.dtype == torch.float32
            assert torch.allclose(time.double(), ch * 10 + panel + torch.arange(n, dtype=torch.float64)[None, :] / n, atol=1e-5)
            assert torch.equal(stats[:, 0:1], ch + panel) and torch.equal(stats[:, 2:3], ch * 3)


def test_single_process_gather_is_identity():
    from quantum_inferno_amd import dist as qdist

    flat = torch.arange(10, dtype=torch.float64)
    assert torch.equal(qdist.gather_reduced(flat)[0], flat)


def test_bench_rank_plumbing_world2(tmp_path):
    """bench.py itself under `torch.distributed.run` with two gloo ranks and stub transforms (--stub 1): the sharding, the
    message buffers, the two-deep pipelined gather, the settle / barrier logic, the max-over-ranks timing and the JSON line
    are the real code; rank 0's last gathered buffers hold every rank's reduced products of the last two steps."""
    import json
    import subprocess

    from quantum_inferno_amd import dist as qdist

    dump = os.path.join(str(tmp_path), "stub.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--stub", "1", "--log2n", "12", "--channels", "3", "--settle-ms", "0", "--stub-dump", dump]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and len(line["config"]["rank_seconds"]) == 2
    assert line["steps"] == 6 and line["scaling"] == "weak" and line["config"]["channels_per_gpu"] == 3
    assert line["config"]["points_per_step"] == 2 * 6 * line["config"]["bands"] * 4096  # whole job: both ranks' channels
    assert abs(line["ms_per_step"] - max(line["config"]["rank_seconds"]) / 6 * 1e3) < 1e-3  # max over ranks
    got = torch.load(dump)
    n_ch, n_b, n, calls = got["n_ch"], got["n_b"], got["n"], got["calls"]
    assert calls == 2 + 2 + 3 + 6  # two set-up calls, warmup, the profiled steps, the timed steps
    for buf in got["gathered"]:
        assert buf.shape == (2, 2 * got["slots"])
    seen = set()
    for buf in got["gathered"]:
        for rank in range(2):
            parts = qdist.unpack_reduced(buf[rank], n_ch, [(n_b, n), (n_b, n)], torch.float32)
            call = int(parts[0][2][0, 0])
            seen.add(call)
            for k, (band, time, stats) in enumerate(parts):
                assert torch.all(band == 1000.0 * rank + 100.0 * k + call) and torch.all(stats == call)
                assert torch.all(time == float(rank + k))
    assert seen == {calls - 2, calls - 1}


def _torchrun(args, timeout=600):
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *args]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_default_multirank_run_is_configs3_world2():
    """`bench.py --gpus N` with no --config (what the driver's scaling run launches): `value` is BASELINE configs[3] -- 64
    records per GPU, order 12, the STFT in the step, one gather of the reduced product -- and the one-record-per-GPU case
    rides along under its own key.  Stub transforms, short records (2^12), two gloo ranks."""
    line = _torchrun(["--steps", "2", "--warmup", "1", "--stub", "1", "--log2n", "12", "--settle-ms", "0"])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["world_size"] == 2 and cfg["backend"].startswith("gloo")
    assert cfg["channels_per_gpu"] == 64 and "configs[3]" in cfg["workload"] and "128 channels" in cfg["workload"]
    assert cfg["points_per_step"] == 2 * 128 * cfg["bands"] * 4096
    assert len(cfg["rank_seconds"]) == 2 and len(cfg["gather_wait_ms_per_step"]) == 2
    assert cfg["gather_message_bytes_per_rank"] > 0
    nested = line["configs1_per_gpu"]
    assert nested["config"]["channels_per_gpu"] == 1 and nested["n_gpus"] == 2 and nested["steps"] == 2
    assert "scaling_note" in line


def test_bench_default_multirank_run_world8():
    """The driver's N = 8 command line as it is (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`), with stub
    transforms and short records on eight gloo ranks: configs[3] = 512 channels, eight rows in every gathered buffer, every
    rank's seconds, waits and stage times in the line, value = the whole job's points over the slowest rank's time."""
    import json
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--stub", "1", "--log2n", "10", "--settle-ms", "0"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["world_size"] == 8 and line["scaling"] == "weak"
    assert cfg["channels_per_gpu"] == 64 and "512 channels" in cfg["workload"] and "configs[3]" in cfg["workload"]
    assert cfg["points_per_step"] == 2 * 512 * cfg["bands"] * 1024
    assert len(cfg["rank_seconds"]) == 8 and len(cfg["gather_wait_ms_per_step"]) == 8
    assert all(len(v) == 8 for v in cfg["rank_stage_ms_per_step"].values())
    assert abs(line["ms_per_step"] - max(cfg["rank_seconds"]) / 3 * 1e3) < 1e-3
    assert abs(line["value"] - cfg["points_per_step"] / (line["ms_per_step"] * 1e-3) / 1e6) <= 1e-4 * line["value"]  # (rounded figures)
    assert line["configs1_per_gpu"]["n_gpus"] == 8


def test_bench_stream_items_sharded_world2(tmp_path):
    """`bench.py --config 4 --stub 1` on two gloo ranks: the (channel block, chunk) items of the record set are dealt by
    stream.rank_items -- every item exactly once, a rank touches only records it owns (OwnedRecords raises otherwise),
    the JSON line counts both ranks' items."""
    from quantum_inferno_amd import stream

    dump = os.path.join(str(tmp_path), "stream")
    line = _torchrun(["--config", "4", "--stub", "1", "--log2n", "12", "--channels", "2", "--stream-chunks", "7", "--warmup", "2",
                      "--stub-dump", dump])
    n, hop, chunks, block = 4096, 2048, 7, 2
    total = n + (chunks - 1) * hop
    items = stream.work_items(2 * block, block, total, n, hop)
    assert len(items) == 2 * chunks
    seen = []
    for rank in range(2):
        got = torch.load(f"{dump}.rank{rank}")
        mine = [tuple(i) for i in got["items"]]
        assert mine == [tuple(i) for i in stream.rank_items(items, rank, 2)]
        assert all(c0 == rank * block for c0, _, _, _ in mine)  # a rank's share is its own block of records
        assert got["timed"] == [(c0, ch) for c0, _, ch, _ in mine[got["warm"]:]]
        seen += mine
    assert sorted(seen) == sorted(tuple(i) for i in items)
    assert line["config"]["rank_items"] == [chunks - 2, chunks - 2] and line["steps"] == chunks - 2
    assert line["config"]["world_size"] == 2 and line["n_gpus"] == 2
    assert line["config"]["points_per_step"] == 2 * block * line["config"]["bands"] * n * 2


def test_bench_scaling_base_consistent_with_multirank_value_world2():
    """The one-GPU point of the weak-scaling curve is machine-readable in the N = 1 line (`scaling_base` = its configs2 record:
    the N = 1 `value` itself is configs[1], another workload) and consistent with the N-rank line: with the stub's fixed step
    time the two-rank `value` is ~2 x `scaling_base.value`, `points_per_step` doubles, both lines carry `scaling_note`, and the
    N-rank line has every rank's stage times."""
    import json
    import subprocess

    common = ["--steps", "2", "--warmup", "1", "--stub", "1", "--stub-ms", "400", "--log2n", "12", "--settle-ms", "0"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--legs", "main,configs2", *common],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    one = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    base = one["scaling_base"]
    assert one["config"]["channels_per_gpu"] == 1 and "configs[1]" in one["config"]["workload"]
    assert base["value"] == one["configs2"]["value"] and base["channels_per_gpu"] == 64 and base["n_gpus"] == 1
    assert "configs[2]" in base["workload"] and "scaling_note" in one
    assert base["points_per_step"] == 2 * 64 * one["configs2"]["config"]["bands"] * 4096
    two = _torchrun(["--legs", "main", *common])
    assert "configs[3]" in two["config"]["workload"] and "scaling_note" in two and "scaling_base" not in two
    assert two["config"]["points_per_step"] == 2 * base["points_per_step"]
    assert abs(two["value"] / base["value"] - 2.0) < 0.3, (two["value"], base["value"])
    assert abs(two["value"] - two["config"]["points_per_step"] * two["steps"] / max(two["config"]["rank_seconds"]) / 1e6) <= 1e-3 * two["value"]
    stages = two["config"]["rank_stage_ms_per_step"]
    assert stages and all(len(v) == 2 for v in stages.values()) and "block" in stages and "zoom" in stages


def test_forced_collective_on_one_rank_gloo():
    """`force_collective` takes a ONE-rank group through the real gather (what tests/test_gpu_rccl.py does with RCCL on the
    GPU box), here with gloo: GatherPipeline with buffer reuse, and bench.py's N-rank branch under --force-collective 1."""
    import json
    import subprocess

    code = (
        "import torch, torch.distributed as dist\n"
        "from quantum_inferno_amd import dist as qdist\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        "base = torch.arange(64, dtype=torch.float64)\n"
        "msgs = [torch.empty(64, dtype=torch.float64) for _ in range(2)]\n"
        "pipe = qdist.GatherPipeline(depth=2, timing=True, force_collective=True)\n"
        "seen = []\n"
        "for k in range(7):\n"
        "    i = pipe.acquire()\n"
        "    if pipe.outs[i] is not None: seen.append((k - 2, pipe.outs[i].clone()))\n"
        "    msgs[i].copy_(base + 10.0 * k)\n"
        "    out = pipe.submit(i, msgs[i])\n"
        "    assert out.data_ptr() != msgs[i].data_ptr()\n"
        "outs = pipe.drain()\n"
        "assert all(torch.equal(b[0], base + 10.0 * k) for k, b in seen) and len(seen) == 5\n"
        "assert torch.equal(outs[0][0], base + 60.0) and torch.equal(outs[1][0], base + 50.0)\n"
        "assert qdist.gather_reduced(base)[0].data_ptr() == base.data_ptr()  # unforced: the message itself\n"
        "dist.destroy_process_group()\n"
    )
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), PYTHONPATH=ROOT)
    run = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr[-3000:]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--channels", "3", "--log2n", "12", "--stub", "1",
           "--force-collective", "1", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0"]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["config"]["backend"] == "gloo" and line["config"]["world_size"] == 1
    assert line["config"]["gather_message_bytes_per_rank"] > 0 and line["config"]["gather_wait_ms_per_step"] is not None
