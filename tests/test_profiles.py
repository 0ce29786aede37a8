"""The committed profile evidence is self-consistent: the table bench.py quotes traffic / instruction figures from
(profiles/traffic.json) holds exactly the figures of the committed PMC summary of the same round - both are written
by ONE run of profiles/summarize_pmc.py (profiles/run_profile.sh)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _profile_round():
    src = open(os.path.join(ROOT, "bench.py")).read()
    return re.search(r'^PROFILE_ROUND = "(\w+)"', src, re.M).group(1)


def _summary_tables(path):
    """kernel -> the '=> table:' figures of a pmc_summary.txt, and the workload key of its 'wrote' line."""
    tables, kernel, workload = {}, None, None
    for line in open(path):
        if line.startswith("# ") and not line.startswith("# kernel stats"):
            kernel = line[2:].strip()
        m = re.match(r"\s+=> table: (\{.*\})", line)
        if m:
            tables[kernel] = json.loads(m.group(1))
        m = re.match(r"wrote \S+ workload (\S+)", line)
        if m:
            workload = m.group(1)
    return tables, workload


def test_traffic_table_equals_the_committed_summary():
    rnd = _profile_round()
    summary = os.path.join(ROOT, "profiles", rnd, "pmc_summary.txt")
    assert os.path.exists(summary), f"bench.py names profiles/{rnd} but {summary} is missing"
    tables, workload = _summary_tables(summary)
    assert workload, "the summary does not say which workload it was written for"
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert workload in traffic
    assert traffic[workload] == tables
    dominant = traffic[workload]["traceKernel<false, false, false, false>"]
    assert dominant["hbm_bytes_per_launch"] == 2 * dominant["fetch_size_bytes_raw"] + dominant["write_size_bytes"]
    assert dominant["valu_insts_per_launch"] > 0


def test_kernel_stats_of_the_round_are_committed():
    rnd = _profile_round()
    stats = os.path.join(ROOT, "profiles", rnd, "trace_kernel_stats.csv")
    assert os.path.exists(stats)
    text = open(stats).read()
    assert "traceKernel<false, false, false, false>" in text and "AverageNs" in text
