"""bench.py prints one JSON line with the driver's contract (plus `roofline` and `cpu_baseline`)."""
import json
import os
import subprocess
import sys

import pytest

import golden_util as gu


@pytest.mark.gpu
def test_bench_tiny_json_contract():
    r = subprocess.run([sys.executable, os.path.join(gu.ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mreads/s" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["parity_with_gpu_on_sample"] is True
    # the CPU baseline states its real resources: CPUs in the affinity mask, the cgroup's CPU-time quota, threads it ran
    for key in ("cores_visible", "cpu_quota", "threads_used"):
        assert key in cb, key
    assert 1 <= cb["threads_used"] <= cb["cores_visible"] and cb["cores"] == cb["threads_used"]
    assert cb["cpu_quota"] is None or cb["threads_used"] <= cb["cpu_quota"] + 1
    assert d["value"] > 0 and abs(d["value"] - d["config"]["reads_per_gpu"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02
    assert d["known_answer"]["label_and_count_ok"] == 1.0
    # the two extra legs (SURVEY.md 8d ii, iii): batch-API pipeline and files-in / CSV-out through exe/cuCLARK
    assert d["pipeline"]["results_equal_device_path"] is True and d["pipeline"]["value"] > 0
    e2e = d["end_to_end"]
    assert "error" not in e2e, e2e
    assert e2e["csv_lines_equal_kernel_rows"] is True and e2e["objects"] == d["config"]["reads_per_gpu"]
    assert e2e["ingest"]["batches_through_host_path"] == 0
    # the headline is the table the command line builds (one strand); the two-strand table is a side leg with equal rows
    assert "both strands" not in d["config"]["table"]["layout"] and d["two_strand_table"]["results_equal_headline_table"] is True
    assert d["skipped_legs"] == [] and d["cut_off_legs"] == [] and d["wall_s"] < d["time_budget_s"]


def _bench(*args, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(gu.ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def _check_config(d, n_reads, random_hits_possible=False):
    assert d["config"]["reads_per_gpu"] == n_reads
    assert d["cpu_baseline"]["parity_with_gpu_on_sample"] is True           # bit-exact vs the oracle on the sample
    ka = d["known_answer"]
    # (k = 27: 1.8e8 stored 27-mers of 1.8e16 - a few of the 2 M random reads do meet one; k = 31: none in 1e12 draws)
    assert ka["label_and_count_ok"] == 1.0 and (ka["random_reads_no_hit"] == 1.0 or (random_hits_possible and ka["random_reads_no_hit"] > 0.99999))
    assert d["config"]["flagged_reads_dense_path"] == 0
    assert d["pipeline"]["results_equal_device_path"] is True
    e2e = d["end_to_end"]
    assert "error" not in e2e, e2e
    assert e2e["csv_lines_equal_kernel_rows"] is True and e2e["objects"] == n_reads
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.02


@pytest.mark.gpu
def test_bench_full_config3():
    """BASELINE.json configs[2]: 10 M x 150 bp against the 36 GB-scale k = 31 table resident in HBM: oracle parity on a
    sample, constructive known answer on all genome reads, batch-API pipeline equal to the device path, CLI CSV equal to
    the kernel's rows on all 10 M reads."""
    d = _bench("--workload", "full", "--steps", "2", "--warmup", "1", "--cpu-sample", "200000", "--no-parts-proxy", "--no-default-layout",
               "--e2e-reps", "1", "--multi-engine-reads", "2000000", "--multi-engine-runs", "db_sharded_2")
    _check_config(d, 10_000_000)
    assert d["config"]["table"]["htsize"] == 1610612741 and d["config"]["table"]["kmers"] > 5_700_000_000
    # the product binary's table-sharded mode against the 36 GB-scale database (every shape: test_bench_light27_config2_proper; all of
    # them at full size: the bench line itself, 8 parts and the 2-D shape included): 2 parts on two engines, 2 M reads, CSV equal
    # to the one-engine run's
    me = d["end_to_end"]["multi_engine"]
    assert me["reads"] == 2_000_000 and me["all_csv_equal"] is True, me
    r4 = me["runs"]["db_sharded_2"]
    assert r4["csv_equals_one_engine_run"] is True and r4["ingest"]["batches_through_host_path"] == 0
    assert "query_kernel_r<31, 20, false, true>" in r4["kernel"] and "2 part(s) x 1 read group(s)" in r4["layout"]
    assert r4["per_batch"]["exchange_MB"] > 0 and r4["per_batch"]["fanout_MB"] > 0 and r4["per_batch"]["kernel_ms_slowest_engine"] > 0


@pytest.mark.gpu
def test_bench_light_config2():
    """BASELINE.json configs[1]: the same 10 M reads against the CuCLARK-l-scale table."""
    d = _bench("--workload", "light", "--steps", "2", "--warmup", "1", "--cpu-sample", "200000", "--no-parts-proxy", "--no-default-layout",
               "--e2e-reps", "1", "--no-multi-engine")
    _check_config(d, 10_000_000)
    assert d["config"]["table"]["htsize"] == 57777779


@pytest.mark.gpu
def test_bench_paired_config5_shape():
    """BASELINE.json configs[4] on one GPU: paired-end 2 x 150 bp objects (read 1 + N + read 2, file.cc:205-268); the
    end-to-end leg feeds the two FASTQ files to exe/cuCLARK -P."""
    d = _bench("--workload", "paired", "--steps", "2", "--warmup", "1", "--cpu-sample", "100000", "--reads", "2000000", "--no-parts-proxy",
               "--no-default-layout", "--e2e-reps", "1", "--no-multi-engine")
    _check_config(d, 2_000_000)
    assert "pairs" in d["end_to_end"]["input"]
    # config 5 names gzip input: plain gzip and block gzip of the same pairs give the plain run's CSV
    gz = d["end_to_end"]["gzip_input"]
    assert gz["pairs"] == 1_000_000
    for kind in ("gzip", "bgzf"):
        assert gz[kind]["csv_equals_plain_run"] is True and gz[kind]["Mpairs_s"] > 0, gz
    assert gz["gzip"]["inflated_on"] == "device" and gz["bgzf"]["inflated_on"] == "device", gz


@pytest.mark.gpu
def test_bench_table_sharded_mode_under_rccl_on_one_gpu():
    """--mode db at N = 1 with the nccl backend: RCCL is initialised and the table-sharded pass runs through its collectives
    (a degenerate all_to_all_single per chunk on device tensors, asynchronous, the overflow all-gather, the in-group gather) -
    the code path the driver launches on eight GPUs, executed once on one."""
    d = _bench("--workload", "tiny", "--mode", "db", "--backend", "nccl", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert d["known_answer"]["label_and_count_ok"] == 1.0 and d["known_answer"]["random_reads_no_hit"] == 1.0
    assert "1 part(s) x 1 read group(s)" in d["config"]["mode"]


@pytest.mark.gpu
def test_bench_light27_config2_proper():
    """SURVEY.md 8d config 2 as cuCLARK-l builds it: HTSIZE 57 777 779, k = 27, u32 keys, ~90 M k-mers, 10 M reads."""
    d = _bench("--workload", "light27", "--steps", "2", "--warmup", "1", "--cpu-sample", "200000", "--no-parts-proxy", "--no-default-layout",
               "--multi-engine-reads", "2000000",
               "--multi-engine-runs", "db_sharded_2,db_sharded_4,db_sharded_8,db_sharded_4_parts_2,read_sharded_2,paired_gzip_read_sharded_2")
    _check_config(d, 10_000_000, random_hits_possible=True)
    # end_to_end: the median of three runs of the command, every run's CSV the same, the stages' busy shares and a named bound
    e2e = d["end_to_end"]
    assert len(e2e["runs"]) == 3 and e2e["min"] <= e2e["value"] <= e2e["max"] and e2e["runs_csv_equal"] is True
    assert e2e["bound"] and set(e2e["stage_busy_share"]) == {"loaders", "device_threads", "writer"} and e2e["loaders_alone_GBs"] > 0
    # every multi-device mode of the product binary on this one GPU (MIC_SHARD_ENGINES), 2 M reads, CSVs equal to the one-engine run's
    me = e2e["multi_engine"]
    assert set(me["runs"]) == {"db_sharded_2", "db_sharded_4", "db_sharded_8", "db_sharded_4_parts_2", "read_sharded_2", "paired_gzip_read_sharded_2"}
    for name, run in me["runs"].items():
        assert "error" not in run and run["csv_equals_one_engine_run"] is True, (name, run)
        if name.startswith("db_sharded"):
            assert run["ingest"]["batches_through_host_path"] == 0 and "query_kernel_r<27, 20, false, true>" in run["kernel"], (name, run)
            assert run["per_batch"]["exchange_MB"] > 0, (name, run)
    assert me["runs"]["paired_gzip_read_sharded_2"]["inflated_on"] == "device"
    t = d["config"]["table"]
    assert t["htsize"] == 57777779 and d["config"]["k"] == 27 and 80_000_000 < t["kmers"] < 95_000_000
    assert "k=27" in d["metric"]


def test_bench_has_the_contract_flags():
    src = open(os.path.join(gu.ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert "oracle" in src and "cpu_baseline" in src and "no_cpu" in src


def _run_ranks(cmd, env, port=None):
    """Runs a multi-rank bench command with the ranks' watchdog on (MIC_BENCH_WATCHDOG: a rank still running after 240 s prints
    every thread's Python stack and exits non-zero).  Nothing is retried: a rank that does not get past its start FAILS the test, and
    what the watchdog printed is kept under gpurun_out/ for the post-mortem."""
    env = dict(env, MIC_BENCH_WATCHDOG="240")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0 and "Timeout (" in r.stderr:
        out_dir = os.path.join(gu.ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "bench_rank_hang.log"), "a") as f:
            f.write(f"--- {' '.join(cmd)}\n{r.stderr[-20000:]}\n")
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["db", "db_2d", "read"])
def test_bench_two_ranks_on_one_gpu(mode):
    """The N>1 code paths of bench.py with two ranks sharing cuda:0 (rows exchanged through host memory with gloo):
    table-sharded mode must reproduce the constructive known answer after exchange + merge + gather."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _run_ranks([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29533", os.path.join(gu.ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                    "--warmup", "1", "--workload", "tiny", "--mode", mode[:2] if mode != "read" else mode, "--backend", "gloo",
                    *(["--parts", "1"] if mode == "db_2d" else [])], env, port=29533)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2
    assert d["scaling"] == ("weak" if mode == "read" else "strong")
    assert d["known_answer"]["label_and_count_ok"] == 1.0 and d["known_answer"]["random_reads_no_hit"] == 1.0
    per = d["config"]["reads_per_gpu"]
    total = per * (2 if mode == "read" else 1)
    assert abs(d["value"] - total / d["ms_per_step"] / 1e3) / d["value"] < 0.02
    if mode == "db":
        assert "2 part(s) x 1 read group(s)" in d["config"]["mode"]
    if mode == "db_2d":      # one part per group: the ranks split the reads, the exchange is degenerate
        assert "1 part(s) x 2 read group(s)" in d["config"]["mode"]
    if mode == "read":   # the extra table-sharded leg rides along and must reproduce the known answer too
        ts = d["table_sharded"]
        assert "error" not in ts, ts
        assert ts["scaling"] == "strong" and ts["value"] > 0
        assert ts["known_answer"]["label_and_count_ok"] == 1.0 and ts["known_answer"]["random_reads_no_hit"] == 1.0
    else:
        assert "table_sharded" not in d


@pytest.mark.gpu
def test_bench_line_stands_when_the_table_sharded_leg_does_not_finish():
    """N > 1, read mode: the extra table-sharded leg has a deadline (--db-leg-timeout); past it rank 0 prints the line without the
    leg and every rank leaves with exit code 0 - a stuck collective in that leg cannot take the headline with it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _run_ranks([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29541", os.path.join(gu.ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                    "--warmup", "1", "--workload", "tiny", "--mode", "read", "--backend", "gloo", "--db-leg-timeout", "0.001"], env, port=29541)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "did not finish" in d["table_sharded"]["error"]


@pytest.mark.gpu
def test_bench_four_ranks_two_dimensional_layout():
    """Four ranks sharing cuda:0 over gloo, read mode: the headline leg, then the table-sharded legs - 4 parts x 1 group (the
    reference's mode) and 2 parts x 2 read groups (the 2-D layout, exchange inside process subgroups) - each must reproduce the
    constructive known answer on all reads."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _run_ranks([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                    "127.0.0.1", "--master-port", "29537", os.path.join(gu.ROOT, "bench.py"), "--gpus", "4", "--steps", "2",
                    "--warmup", "1", "--workload", "tiny", "--mode", "read", "--backend", "gloo"], env, port=29537)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["scaling"] == "weak"
    ts = d["table_sharded"]
    assert "error" not in ts, ts
    assert ts["parts"] == 4 and ts["read_groups"] == 1 and ts["known_answer"]["label_and_count_ok"] == 1.0
    td = ts["two_parts_2d"]
    assert td["parts"] == 2 and td["read_groups"] == 2 and td["reads_this_rank"] == d["config"]["reads_per_gpu"] // 2
    assert td["known_answer"]["label_and_count_ok"] == 1.0 and td["known_answer"]["random_reads_no_hit"] == 1.0


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the script starts its ranks as a child process under
    torch.distributed.run before it touches the GPU and relays rank 0's line (gloo: the two ranks share this box's one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = _run_ranks([sys.executable, os.path.join(gu.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "tiny",
                    "--backend", "gloo", "--no-db-leg"], env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["known_answer"]["label_and_count_ok"] == 1.0
    assert abs(d["value"] - 2 * d["config"]["reads_per_gpu"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02


def _refusal(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(gu.ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)


def test_bench_refuses_more_rccl_ranks_than_devices():
    """--gpus N over RCCL with fewer visible devices than ranks exits non-zero with the reason (no rank is stacked on another's GPU);
    runs anywhere: the check comes before anything touches a GPU (here: no device at all, or one)."""
    import torch
    n = torch.cuda.device_count()
    r = _refusal(["--gpus", str(n + 2), "--workload", "tiny", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "RCCL needs one device per rank" in r.stderr and f"has {n}" in r.stderr, r.stderr[-1000:]
    # the same inside a rank a launcher started (WORLD_SIZE set): a local rank without a device of its own refuses
    r = _refusal(["--gpus", str(n + 2), "--workload", "tiny"], {"WORLD_SIZE": str(n + 2), "RANK": str(n + 1), "LOCAL_RANK": str(n + 1)})
    assert r.returncode != 0 and "has no GPU of its own" in r.stderr, r.stderr[-1000:]
    # a launcher that started another number of ranks than --gpus says
    r = _refusal(["--gpus", "4", "--workload", "tiny"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_bench_fragmented_database_known_answer():
    """A database of discriminative k-mers (mic_synth_spec.keep_ppm: half of the genomes' k-mer positions kept, in runs of geometric
    length - what the removal of common k-mers leaves, HashTableStorage_hh.hh:241-292): oracle parity on the sample, the constructive
    known answer counts the KEPT windows only, the one-strand and the two-strand table agree on every read."""
    d = _bench("--workload", "tiny_frag", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-pipeline", "--no-parts-proxy")
    assert d["cpu_baseline"]["parity_with_gpu_on_sample"] is True
    ka = d["known_answer"]
    assert ka["label_and_count_ok"] == 1.0 and ka["random_reads_no_hit"] == 1.0
    assert 0.3 < d["config"]["table"]["kmers"] / 1_500_000 < 0.7
    assert d["two_strand_table"]["results_equal_headline_table"] is True


@pytest.mark.gpu
def test_bench_time_budget_skips_legs_and_names_them():
    """--time-budget: extra legs that would start past it are skipped and named in the line; every headline field is there."""
    d = _bench("--workload", "tiny", "--steps", "2", "--warmup", "1", "--time-budget", "1")
    for key in ("metric", "value", "roofline", "cpu_baseline", "known_answer"):
        assert key in d, key
    assert d["cpu_baseline"]["parity_with_gpu_on_sample"] is True
    assert {"pipeline", "table_sharded_proxy", "two_strand_table", "end_to_end"} <= set(d["skipped_legs"]), d["skipped_legs"]
    assert "pipeline" not in d and "end_to_end" not in d


@pytest.mark.gpu
def test_bench_tandem_repeats_workload():
    """Genomes with 5 % tandem repeats, the short units shared by all genomes (mic_synth_spec.repeat_ppm): crowded minimizers in a
    side table, reads through crowd_finish_kernel; oracle parity on the sample, known answer on the reads that touch no repeat,
    one-strand, two-strand and direct tables equal on every read."""
    d = _bench("--workload", "tiny_repeats", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-pipeline", "--no-parts-proxy", "--cross-layouts", "direct")
    assert d["cpu_baseline"]["parity_with_gpu_on_sample"] is True
    ka = d["known_answer"]
    assert ka["label_and_count_ok"] == 1.0 and ka["random_reads_no_hit"] == 1.0
    assert d["config"]["crowded"]["side_table_kmers"] > 0 and d["config"]["crowded"]["reads_with_crowded_runs"] > 0, d["config"]["crowded"]
    assert d["two_strand_table"]["results_equal_headline_table"] is True and d["cross_layouts"]["all_equal"] is True


@pytest.mark.gpu
def test_bench_mosaic_labels_workload_ties_and_dense_rows():
    """15 % of every genome in segments whose k-mers' labels change every 1 / 2 / 4 / 8 positions (mic_synth_spec.mosaic_ppm): ties
    between best and second (lower target wins, CuClarkDB.cu:1445-1457), reads with more than 64 targets (the exact dense recount);
    oracle parity on the sample, all layouts equal on every read."""
    d = _bench("--workload", "tiny_homolog", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-pipeline", "--no-parts-proxy", "--cross-layouts", "direct,minimizer")
    assert d["cpu_baseline"]["parity_with_gpu_on_sample"] is True
    assert d["known_answer"]["tie_rate"] > 0.01 and d["config"]["flagged_reads_dense_path"] > 0, (d["known_answer"], d["config"]["flagged_reads_dense_path"])
    assert d["two_strand_table"]["results_equal_headline_table"] is True and d["cross_layouts"]["all_equal"] is True
