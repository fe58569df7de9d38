"""CPU, world_size 2: the multi-GPU path (contiguous row shards per rank, results delivered to rank 0) is exercised
without GPUs -- through the TSV task's OWN code (run_tsv_inference), which must form the process group itself
(gloo here, RCCL on a GPU box) or fall back to the reference's shard-file poll + concat (inference.py:214-225)."""
import base64
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_ROWS = 11


def _write_inputs(tmpdir, with_questions):
    sys.path.insert(0, ROOT)
    from generativeimage2text_amd import tsv_io
    rows = [["key%d" % i, base64.b64encode(b"image-bytes-%03d" % i).decode()] for i in range(N_ROWS)]
    tsv_io.tsv_writer(rows, os.path.join(tmpdir, "img.tsv"))
    if with_questions:
        q = [["key%d" % i, json.dumps([{"question": "q%d-%d" % (i, j), "question_id": 100 * i + j} for j in range(1 + i % 3)])]
             for i in range(N_ROWS)]
        tsv_io.tsv_writer(q, os.path.join(tmpdir, "q.tsv"))


def _task_worker(rank, world, port, tmpdir, with_questions, rendezvous):
    """What one rank of `torchrun`/`mpirun` runs: ONLY environment variables, no dist calls of its own."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for k in ("MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    if rendezvous:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from generativeimage2text_amd import inference
    assert not dist.is_initialized()
    inference.run_tsv_inference(
        os.path.join(tmpdir, "img.tsv"), os.path.join(tmpdir, "q.tsv") if with_questions else None,
        os.path.join(tmpdir, "out.tsv"),
        transform=lambda b: b.decode(),                                         # the "image" is its byte string
        caption_batch=lambda imgs: ["cap<%s>" % im for im in imgs],
        answer_questions=lambda img, qs: ["ans<%s|%s>" % (img, q) for q in qs],
        batch_size=4, poll_s=0.05)
    assert dist.is_initialized() == rendezvous
    if rendezvous:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("rendezvous", [True, False], ids=["gather", "shard-file-fallback"])
@pytest.mark.parametrize("with_questions", [False, True], ids=["caption", "vqa"])
def test_two_rank_tsv_task_delivers_all_rows_on_rank0(tmp_path, with_questions, rendezvous):
    from generativeimage2text_amd import tsv_io, inference
    _write_inputs(str(tmp_path), with_questions)
    mp.spawn(_task_worker, args=(2, _free_port(), str(tmp_path), with_questions, rendezvous), nprocs=2, join=True)
    out = str(tmp_path / "out.tsv")
    rows = list(tsv_io.tsv_reader(out))
    if with_questions:
        want = []
        for i in range(N_ROWS):
            for j in range(1 + i % 3):
                want.append([inference.json_dump({"answer": "ans<image-bytes-%03d|q%d-%d>" % (i, i, j), "question_id": 100 * i + j})])
        assert rows == want
        # the reference's convert_tsv_to_vqa_json reads exactly one column per row (inference.py:227-229)
        assert [json.loads(s)["question_id"] for s, in tsv_io.tsv_reader(out)] == [json.loads(w[0])["question_id"] for w in want]
    else:
        assert rows == [["key%d" % i, inference.json_dump([{"caption": "cap<image-bytes-%03d>" % i}])] for i in range(N_ROWS)]
    # both shard files exist, and the line index of the merged file addresses every row
    for r in range(2):
        assert os.path.isfile("%s.%d.2.tsv" % (out, r))
    t = tsv_io.TSVFile(out)
    assert len(t) == len(rows) and t[len(rows) - 1] == rows[-1]


def _bench_gather_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # bench.py's token gather: every rank contributes [B, T] tokens + [B] logprobs
    toks = torch.full((3, 5), rank, dtype=torch.int64)
    lps = torch.full((3,), float(rank))
    g_t, g_l = bench.gather_results(toks, lps)
    if rank == 0:
        assert g_t.shape == (world * 3, 5) and g_t[3:].eq(1).all() and g_t[:3].eq(0).all()
        assert g_l.tolist() == [0.0] * 3 + [1.0] * 3
        open(os.path.join(tmpdir, "ok"), "w").write("1")
    else:
        assert g_t is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bench_gather(tmp_path):
    mp.spawn(_bench_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


class _StandInEngine:
    """What bench.main needs from an engine context: generate() returning (tokens [B, T], logprobs [B], info [4])."""

    def __init__(self, rank, batch, T):
        self.rank, self.batch, self.T = rank, batch, T
        self.calls = 0
        self.log = []

    def clone(self):
        c = _StandInEngine(self.rank, self.batch, self.T)
        c.log = self.log
        return c

    def set_encode_after(self, other):
        pass

    def set_graph(self, on):
        pass

    def generate(self, frames, search, sync=True):
        self.calls += 1
        toks = torch.full((self.batch, self.T), 1000 + self.rank, dtype=torch.int64)
        return toks, torch.full((self.batch,), -float(self.rank)), torch.tensor([self.T, 0, self.T - 1, 0], dtype=torch.int32)


def _bench_main_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    gathered = []
    real_gather = bench.gather_results

    def spy(tokens, logprobs):
        out = real_gather(tokens, logprobs)
        gathered.append(out)
        return out
    bench.gather_results = spy
    res = bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--batch", "3", "--contexts", "2",
                      "--no-cpu-baseline"],
                     engine_factory=lambda args, r: (_StandInEngine(r, args.batch, args.max_steps), [torch.zeros(1)]))
    if rank == 0:
        assert res["n_gpus"] == world and res["config"]["global_batch"] == world * 3 and res["config"]["parallelism"] == f"dp{world}"
        assert res["value"] > 0 and res["steps"] == 4
        t, l = gathered[-1]                                   # rank 0 received the rows of EVERY rank, in rank order
        assert t.shape == (world * 3, 20)
        for r in range(world):
            assert t[3 * r:3 * r + 3].eq(1000 + r).all() and l[3 * r:3 * r + 3].tolist() == [-float(r)] * 3
        # the path's one collective is reported on its own: in the schedule and alone
        g = res["gather"]
        assert g["bytes_per_rank"] == 3 * 21 * 8 and g["alone_ms"]["median"] > 0 and g["in_schedule_ms"]["median"] >= 0
        open(os.path.join(tmpdir, "ok"), "w").write(json.dumps(res))
    else:
        assert res is None and gathered[-1] == (None, None)


def test_two_rank_bench_main_with_standin_engine(tmp_path):
    """bench.main under a 2-rank launcher environment (gloo, stand-in engine): every rank runs, rank 0 prints ONE line
    with n_gpus == --gpus == WORLD_SIZE, the gather delivers every rank's rows to rank 0."""
    mp.spawn(_bench_main_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert json.loads((tmp_path / "ok").read_text())["n_gpus"] == 2


def test_eight_rank_bench_main_with_standin_engine(tmp_path):
    """The same at the node's size: 8 ranks (BASELINE.json configs[3]: DP across 8 GPUs).  Every rank times its steps, the
    elapsed time is the MAX over ranks, rank 0 receives 8 x B rows in rank order and prints ONE line with n_gpus = 8,
    global_batch = 8 B, parallelism dp8 and the gather figures."""
    mp.spawn(_bench_main_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    res = json.loads((tmp_path / "ok").read_text())
    assert res["n_gpus"] == 8 and res["config"]["global_batch"] == 24 and res["scaling"] == "weak"


def test_bench_refuses_gpu_count_mismatch(tmp_path):
    """`python bench.py --gpus 2` must not print a 1-GPU line: with a launcher environment that disagrees it exits, and
    without one it starts the ranks itself -- which fails loudly here because fewer than 2 GPUs are visible."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"metric"' not in r.stdout


def test_json_dump_is_the_reference_format():
    from generativeimage2text_amd import inference
    assert inference.json_dump({"question_id": 7, "answer": "a b"}) == '{"answer":"a b","question_id":7}'
    assert inference.json_dump([{"caption": "x"}]) == '[{"caption":"x"}]'
