#!/usr/bin/env python3
"""One-screen digest of gpurun_out/measure/ (tools/measure_round.sh): the figures DESIGN.md section 5 quotes."""
import json, sys
M = (sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/measure') + '/'
d = json.load(open(M + 'bench_n1_full.json'))
print('printed line:', len(open(M + 'bench_n1.json').read().strip()), 'characters')
print('headline', round(d['value'], 1), 'scans/s', round(d['ms_per_step'], 2), 'ms/step, iterations', d['mean_iterations'], 'set_map', round(d['set_map_ms'], 2),
      'ms, fixed-30', round(d['fixed_30_iterations']['scans_per_s'], 1))
h = d['host_input']
print('  host input: pinned', round(h['pinned_over_device_resident'], 3), h['pinned_ms_each_step']['step_ms'], 'pageable', round(h['pageable_over_device_resident'], 3),
      h['pageable_ms_each_step']['step_ms'])
r = d['roofline']
print('  roofline: frac', round(r['frac'], 4), 'launch', round(r['avg_launch_us'], 1), 'us, shared-map frac', round(r['frac_shared_map'], 4), 'traffic', r['traffic'],
      'active problems', round(r['active_problems_per_launch'], 2))
c = d['cpu_baseline']
print('  cpu:', round(c['value'], 2), 'scans/s on', c['cores'], 'cores, single core', round(c['single_core_scans_per_s'], 2), '; GPU/CPU', round(d['speedup_vs_cpu_baseline'], 1),
      round(c['gpu_over_single_core'], 1))
print('  kernels (us per launch):', {k: v['avg_us'] for k, v in d['kernels'].items()})
l = json.load(open(M + 'bench_loopclosure_full.json'))
print('loop closing', round(l['value'], 1), 'pairs/s', round(l['ms_per_step'], 1), 'ms/step, frac', round(l['roofline']['frac'], 4), 'launch', round(l['roofline']['avg_launch_us'], 1),
      'us, cpu', round(l['cpu_baseline']['value'], 2), 'pairs/s on', l['cpu_baseline']['cores'])
for f in ('bench_stream_1', 'bench_stream_4', 'bench_stream_fleet16'):
    x = json.load(open(M + f + '_full.json'))
    print(f, round(x['value'], 1), 'scans/s,', round(x['ms_per_scan_per_vehicle'], 3), 'ms per scan and vehicle, iterations', x['mean_iterations'], 'end error', round(x['final_position_error_m'], 4),
          'frac', x['roofline'] and round(x['roofline']['frac'], 4), 'launch', x['roofline'] and round(x['roofline']['avg_launch_us'], 1), 'cpu', x['cpu_baseline'] and (round(x['cpu_baseline']['value'], 2), x['cpu_baseline']['cores']))
s = json.load(open(M + 'bench_slam_full.json'))
print('slam', round(s['value'], 1), 'scans/s', {k: s['slam'][k] for k in ('keyframes', 'loops_closed', 'map_rebuilds', 'mean_icp_iterations', 'optimizer_host_s', 'localizer_host_s')})
print('  replay', s['replay_vs_oracle'], 'cpu', round(s['cpu_baseline']['value'], 1), s['cpu_baseline']['unit'])
try:
    s1 = json.load(open(M + 'bench_slam100k_full.json'))
    print('slam 100k-pt scans', round(s1['value'], 1), 'scans/s', {k: s1['slam'].get(k) for k in ('keyframes', 'map_rebuilds', 'device_map_rebuilds', 'device_input_stages', 'localizer_host_s', 'input_filters')})
    f = json.load(open(M + 'bench_f64_full.json'))
    print('f64', round(f['value'], 1), 'scans/s', round(f['ms_per_step'], 2), 'ms/step, frac', round(f['roofline']['frac'], 4), 'launch', round(f['roofline']['avg_launch_us'], 1), 'us; x of f32:', round(f['value'] / d['value'], 3))
    print('pmc', {k: v for k, v in json.load(open(M + 'pmc/knn_pmc.json')).items() if k != 'per_launch' and k != 'how' and k != 'note'})
    for leg in ('knn_traffic', 'knn_traffic_loopclosure', 'knn_traffic_stream', 'knn_traffic_f64'):
        t = json.load(open(M + 'pmc/' + leg + '.json'))
        print(' ', leg, 'bytes per launch', round(t['hbm_bytes_per_launch'] / 1e6, 2), 'MB (uncorrected', round(t['hbm_bytes_per_launch_uncorrected'] / 1e6, 2), 'MB), launches', t['launches'])
except Exception as e:
    print('round-4 legs:', type(e).__name__, e)
print('slam mt', open(M + 'slam_mt.json').read()[:400])
n = json.load(open(M + 'bench_normals.json'))
print('normals', {k: (round(v['kernel_ms'], 2) if isinstance(v, dict) else round(v, 1)) for k, v in n.items()})
print(open(M + 'trace_summary.txt').read()[:700])
print(open(M + 'host_input_overlap.txt').read())
