"""Copy the summaries of scripts/r6_final.sh (gpurun_out/r6final/) into profiles/r06_* and write r06_service_launches.txt."""
import csv, gzip, json, shutil
S, P = 'gpurun_out/r6final/', 'profiles/'
for a, b in (('bench_default.json', 'r06_bench_default.json'), ('bench_default_traced.json', 'r06_bench_default_traced.json'),
             ('kernel_stats_bench_default.csv', 'r06_kernel_stats_bench_default.csv'),
             ('kernel_trace_bench_default.csv.gz', 'r06_kernel_trace_bench_default.csv.gz'),
             ('pmc_sq_ring.csv', 'r06_pmc_sq_ring_both_services.csv'), ('pmc_fetch_size_ring.csv', 'r06_pmc_fetch_size_ring_both_services.csv'),
             ('pmc_write_size_ring.csv', 'r06_pmc_write_size_ring_both_services.csv'), ('traffic.json', 'r06_traffic.json'),
             ('bench_force_dist_nccl_1rank.json', 'r06_bench_force_dist_nccl_1rank.json'),
             ('bench_gloo_2ranks_one_gpu.json', 'r06_bench_gloo_2ranks_one_gpu.json')):
    shutil.copy(S + a, P + b)
open(P + 'r06_ring_pmc_runs.jsonl', 'w').write(''.join(open(S + n).read() for n in ('ring_pmc_sq.json', 'ring_pmc_fetch.json', 'ring_pmc_write.json')))
open(P + 'r06_pytest_gpu_tail.log', 'w').write(''.join(open(S + 'pytest.log', errors='replace').readlines()[-12:]))
rows = []
with gzip.open(S + 'kernel_trace_bench_default.csv.gz', 'rt') as f:
    rows = [r for r in csv.DictReader(f) if 'k_sha256' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
d = [json.loads(l) for l in open(S + 'bench_default_traced.json') if l.startswith('{')][0]
r = d['roofline']
stats = [x for x in csv.DictReader(open(S + 'kernel_stats_bench_default.csv')) if 'k_sha256_pair' in x['Name']][0]
what = ['warm-up (5 files)'] * 2 + ['TIMED region (20 files)'] * 2 + ['one file alone'] * 2
out = ["# per-launch durations of the two persistent SHA-256 services in profiles/r06_kernel_trace_bench_default.csv.gz",
       "# (rocprofv3 --kernel-trace of `python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras`, final code of round 6).",
       "# The --stats average (r06_kernel_stats_bench_default.csv: %.1f ms over 3 launches) mixes the warm-up, the TIMED region and the" % (float(stats['AverageNs']) / 1e6),
       "# one-file pass; the bench line's roofline uses the timed launch, measured with HIP events on the service's own stream.",
       "# kernel, start (ms from the first service launch), duration (ms), what it is"]
timed = None
for row, w in zip(rows, what):
    n = 'k_sha256_xpair<RingSource>' if 'xpair' in row['Kernel_Name'] else 'k_sha256_pair<RingSource,false>'
    dur = int(row['DurationNs']) / 1e6
    if w.startswith('TIMED') and 'xpair' not in n:
        timed = dur
    out.append(f"{n:<34}{(int(row['Start_Timestamp']) - t0) / 1e6:>10.3f} {dur:>10.3f}   {w}")
alg = r['algorithmic_bytes_per_launch']
out.append(f"# timed launch by rocprofv3: {timed:.3f} ms; by HIP events in the same run's bench line (roofline.service_launch_ms): {r['service_launch_ms']} ms;")
out.append(f"# algorithmic bytes {alg} / {timed:.3f} ms = {alg / timed / 1e6:.1f} GB/s = {alg / timed / 1e6 / 8000:.4f} of 8 000 GB/s (line: {r['achieved']} GB/s, frac {r['frac']})")
open(P + 'r06_service_launches.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-8:]))
for n in ('bench_default', 'bench_default_traced'):
    d = [json.loads(l) for l in open(S + n + '.json') if l.startswith('{')][0]
    r = d['roofline']
    print(n, d['value'], d['ms_per_step'], r['frac'], r['valu']['frac'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file']['ms'], r['cu_time_budget'].get('chip_ceiling_GiBps'), r['traffic'])
