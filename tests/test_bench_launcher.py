"""bench.py's rank launcher and JSON contract, on CPU (gloo, stubbed step): `--gpus N` must really run N ranks, rank 0 must
print exactly one JSON line that says so, and asking for GPUs that are not there must fail loudly.
Replaces the reference's nn.DataParallel wrapping (utils/model_save_util.py:50-54) as the way N GPUs are driven."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return e


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_launcher_spawns_n_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and len(d["per_rank_ms_per_step"]) == 2
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert abs(d["ms_per_step"] - max(d["per_rank_ms_per_step"])) < 1e-9        # MAX over ranks
    for k in ("metric", "value", "unit", "vs_baseline", "dtype", "data", "config"):
        assert k in d


def test_two_rank_training_line_carries_every_scaling_field():
    """The first N > 1 run on GPUs has to be informative by itself (no 1 -> 8 curve has ever been measured): the training line of a
    2-rank launch must carry ranks_seen, the per-rank step times, `mode`, and the all-reduce object with bytes per step, exposed
    and stand-alone times and where the exchange ran -- checked on the stub step over gloo, image and video mode."""
    import bench
    for mode in ("train", "train_video"):
        p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--mode", mode],
                           env=_env(), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        (d,) = _json_lines(p.stdout)
        for k in bench.TRAIN_LINE_RANK_FIELDS:
            assert k in d, (mode, k)
        assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and len(d["per_rank_ms_per_step"]) == 2
        for k in bench.ALLREDUCE_FIELDS:
            assert k in d["allreduce"], (mode, k)
        assert d["allreduce"]["world"] == 2 and d["allreduce"]["bytes_per_step"] > 0
        assert "data-parallel x2" in d["config"]["parallelism"]
    # and the real training line is assembled from the same field names (bench.train_numbers / train_bench)
    src = open(BENCH).read()
    for k in bench.ALLREDUCE_FIELDS:
        assert ('"%s"' % k) in src.split("def train_numbers")[1].split("def train_bench")[0], k


def test_single_rank_line_has_no_rank_fields():
    p = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--stub"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    (d,) = _json_lines(p.stdout)
    assert d["n_gpus"] == 1 and "ranks_seen" not in d


def test_torchrun_style_ranks_from_environment():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), BENCH, "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--stub"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    (d,) = _json_lines(p.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1]


def test_more_gpus_than_visible_is_refused():
    """No silent single-rank run: the container has no GPU, so --gpus 2 without --stub must exit non-zero with a message."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2
    assert "refusing" in p.stderr and not _json_lines(p.stdout)


def test_visible_gpu_count_reads_sysfs_not_the_runtime(tmp_path, monkeypatch):
    """bench.visible_gpu_count: KFD topology nodes with SIMDs, cut down by the *_VISIBLE_DEVICES lists -- the launcher parent
    must be able to refuse `--gpus N` without initialising HIP."""
    import builtins
    import bench
    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 256, 256, 256]):              # two CPU nodes, three GPUs
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    real_listdir, real_open = os.listdir, builtins.open
    base = "/sys/class/kfd/kfd/topology/nodes"
    monkeypatch.setattr(os, "listdir", lambda p: real_listdir(str(nodes)) if p == base else real_listdir(p))
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace(base, str(nodes)), *a, **k))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count() == 1


def test_one_process_per_leg_retries_a_dead_leg_and_reports_it(tmp_path, monkeypatch, capsys):
    """bench.run_legs: every leg of the default single-GPU line is a child of its own; a leg that dies is run once more and the
    failed attempt is reported in `leg_failures`; a leg that dies twice leaves an error entry, the line is still printed."""
    import json
    import sys
    import bench
    marker = tmp_path / "first_attempt_done"
    child = (
        "import json, os, sys\n"
        "leg = sys.argv[sys.argv.index('--leg') + 1]\n"
        "if leg == 'train_step' and not os.path.exists(%r):\n"
        "    open(%r, 'w').close(); sys.stderr.write('Memory access fault by GPU node-2\\n'); os._exit(134)\n"
        "if leg == 'train_video_step':\n"
        "    sys.exit(3)\n"
        "print('banner line')\n"
        "print(json.dumps({'metric': 'm', 'value': 1.0} if leg == 'forward' else {'ms_per_step': 8.0, 'leg': leg}))\n"
    ) % (str(marker), str(marker))
    monkeypatch.setattr(bench, "_leg_command", lambda argv, leg: [sys.executable, "-c", child] + list(argv) + ["--leg", leg])
    assert bench.run_legs(["--steps", "2"]) == 0
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["metric"] == "m" and line["train_step"] == {"ms_per_step": 8.0, "leg": "train_step"}
    assert "error" in line["train_video_step"]
    fails = line["leg_failures"]
    assert [(f["leg"], f["attempt"], f["rc"]) for f in fails] == [("train_step", 1, 134), ("train_video_step", 1, 3),
                                                                   ("train_video_step", 2, 3)]
    assert "Memory access fault" in fails[0]["stderr_tail"][-1]


def test_leg_children_are_not_used_under_a_launcher_or_for_single_legs(monkeypatch):
    import bench
    called = []
    monkeypatch.setattr(bench, "run_legs", lambda argv: called.append(argv) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main([]) == 0 and called == [[]]
    # --no-train (profiling runs), a training mode of its own, another dtype and the CPU stub stay in this process
    for argv in (["--stub"],):
        called.clear()
        assert bench.main(argv) == 0 and called == []


def test_rank_processes_run_the_training_legs_in_children_first(monkeypatch):
    """under a launcher every rank starts one child per training leg (shifted MASTER_PORT) before its own headline region; a leg
    that fails leaves an error entry and a leg_failures record, the other leg's numbers are kept"""
    import sys
    import bench
    seen = []
    child = (
        "import json, os, sys\n"
        "leg = sys.argv[sys.argv.index('--leg') + 1]\n"
        "sys.stderr.write('port %s\\n' % os.environ['MASTER_PORT'])\n"
        "if leg == 'train_video_step':\n"
        "    sys.exit(7)\n"
        "if os.environ['RANK'] == '0':\n"
        "    print(json.dumps({'ms_per_step': 8.0, 'port': os.environ['MASTER_PORT']}))\n"
    )
    monkeypatch.setattr(bench, "_leg_command", lambda argv, leg: seen.append(leg) or [sys.executable, "-c", child, "--leg", leg])
    monkeypatch.setenv("MASTER_PORT", "29000")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    got, fails = bench.run_rank_training_legs([])
    assert seen == ["train_step", "train_video_step"]
    assert got["train_step"] == {"ms_per_step": 8.0, "port": "29017"} and "error" in got["train_video_step"]
    assert [(f["leg"], f["rc"]) for f in fails] == [("train_video_step", 7)] and fails[0]["stderr_tail"] == ["port 29020"]
    # a rank other than 0 prints nothing: an empty result is not a failure there
    monkeypatch.setenv("RANK", "1")
    got, fails = bench.run_rank_training_legs([], legs=("train_step",))
    assert got == {"train_step": {}} and fails == []


def test_leg_children_of_torchrun_ranks_form_their_own_group(tmp_path):
    """two ranks under `python -m torch.distributed.run` (whose workers are clients of the agent's store) each start the leg
    children; the children must rendezvous among themselves on the shifted port (gloo here) and rank 0's child reports"""
    import subprocess
    import sys
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "import torch, torch.distributed as td\n"
        "td.init_process_group('gloo')\n"
        "t = torch.tensor([float(td.get_rank() + 1)])\n"
        "td.all_reduce(t)\n"
        "if td.get_rank() == 0:\n"
        "    print(json.dumps({'sum': t.item(), 'port': os.environ['MASTER_PORT'], 'leg': sys.argv[sys.argv.index('--leg') + 1]}))\n"
        "td.destroy_process_group()\n")
    parent = tmp_path / "parent.py"
    parent.write_text(
        "import json, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench._leg_command = lambda argv, leg: [sys.executable, %r, '--leg', leg]\n"
        "got, fails = bench.run_rank_training_legs([])\n"
        "open(os.path.join(%r, 'rank%%s.json' %% os.environ['RANK']), 'w').write(json.dumps([got, fails]))\n"
        % (ROOT, str(child), str(tmp_path)))
    port = 29000 + (os.getpid() % 500)
    env = dict(os.environ, UNCL_BENCH_LEG_TIMEOUT="60")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(parent)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=240)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    import json
    got0, fails0 = json.loads((tmp_path / "rank0.json").read_text())
    got1, fails1 = json.loads((tmp_path / "rank1.json").read_text())
    assert fails0 == [] and fails1 == [], (fails0, fails1)
    assert got0["train_step"] == {"sum": 3.0, "port": str(port + 17), "leg": "train_step"}
    assert got0["train_video_step"]["port"] == str(port + 20) and got1 == {"train_step": {}, "train_video_step": {}}


def test_fault_classification_against_a_memory_map(tmp_path):
    """bench.classify_fault: the faulting address of a dead training leg against the allocator table that leg wrote before its
    timed steps (bench.dump_memory_map): inside a live block, inside a freed block, just past a segment, nowhere near."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    mm = tmp_path / "uncl_memmap_train_step_1.json"
    seg = {"address": 0x7f0000000000, "size": 4 << 20, "pool": [0, 0], "stream": 0,
           "blocks": [[0, 1 << 20, "active_allocated"], [1 << 20, 1 << 20, "inactive"], [2 << 20, 2 << 20, "active_allocated"]]}
    mm.write_text(json.dumps({"pid": 1, "tag": "train_step", "segments": [seg]}))
    head = "[bench] memory map of leg train_step: %s (1 segments)\n" % mm

    def fault(addr):
        return bench.classify_fault(head + "Memory access fault by GPU node-2 (Agent handle: 0x1) on address %s. Reason: Page not present.\n"
                                    % hex(addr))
    assert "active_allocated" in fault(0x7f0000000000 + 4096)["where"]
    assert "inactive" in fault(0x7f0000000000 + (1 << 20) + 8192)["where"]
    assert "outside the nearest torch segment" in fault(0x7f0000000000 + (4 << 20) + 4096)["where"]
    assert "not allocator memory" in fault(0x100000)["where"]
    assert bench.classify_fault("no fault here") is None
    assert "no memory map" in bench.classify_fault("Memory access fault by GPU node-2 on address 0x1000. Reason: x")["where"]


def test_layer_table_sums_to_the_surveys_gflop_per_tile():
    """roofline.layers[*].gflop is SURVEY.md 2.3A / 8(d)'s algorithmic count: transposed 3x3 layers over their INPUT pixels (4682.0 MFLOP
    for up_path.3.conv.conv, not the 4756.6 the implicit GEMM multiplies), inc.conv.conv and outc.conv listed though they never
    launch on their own -- so that the column sums to the 18.2858 GFLOP per tile the whole-forward fraction is quoted on."""
    sys.path.insert(0, ROOT)
    import bench
    rows = bench.layer_gflops(200)
    assert len(rows) == 28 and len({r[0] for r in rows}) == 28
    assert abs(sum(r[1] for r in rows) - 200 * bench.GFLOP_PER_TILE) < 0.1          # 3657.2 GFLOP at 200 tiles
    by = {r[0]: r for r in rows}
    assert abs(by["up_path.3.conv.conv"][1] / 200 - 4.6820) < 1e-4 and abs(by["up_path.3.conv.conv"][2] / 200 - 4.7566) < 1e-4
    assert abs(by["inc.conv.conv1"][1] / 200 - 1.1705) < 1e-4 and by["inc.conv.conv1"][1] == by["inc.conv.conv1"][2]
    assert abs(by["down_path.3.mpconv.1.conv1"][1] / 200 - 0.1180) < 1e-4
    assert all(r[2] >= r[1] for r in rows)
    assert abs(bench.DOM_GFLOP_PER_TILE - (by["up_path.3.conv.conv"][1] + by["up_path.3.up"][1]) / 200) < 2e-4
