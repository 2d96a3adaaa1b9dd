"""python tools/publish_profiles.py <rNN>: copy the outputs of tools/run_profiles.sh <rNN> (gpurun_out/<rNN>) into profiles/ and
refresh profiles/traffic.json."""
import csv, json, os, shutil, sys
RND = sys.argv[1] if len(sys.argv) > 1 else 'r05'
NUM = int(RND.lstrip('r'))
src = f'gpurun_out/{RND}'
shutil.copy(f'{src}/bench_line.json', f'profiles/{RND}_bench_line.json')
shutil.copy(f'{src}/stats/b_kernel_stats.csv', f'profiles/{RND}_bench_kernel_stats.csv')
shutil.copy(f'{src}/stats_headline/b_kernel_stats.csv', f'profiles/{RND}_headline_kernel_stats.csv')
shutil.copy(f'{src}/summary.txt', f'profiles/{RND}_paths_pmc_summary.json')
shutil.copy(f'{src}/paths_pmc.json', 'profiles/paths_pmc.json')
line = [l for l in open(f'{src}/stats_headline.log') if l.startswith('{')][-1]
open(f'profiles/{RND}_headline_profiled_line.json', 'w').write(line)
h = json.load(open(f'{src}/pmc_headline.json'))
flat = {k: v['avg_per_dispatch'] for k, v in h.items()}
flat['dispatches'] = h['FETCH_SIZE']['dispatches']
rd, wr = 2 * flat['FETCH_SIZE'] * 1024, flat['WRITE_SIZE'] * 1024
flat['fabric_read_bytes_per_launch'], flat['fabric_write_bytes_per_launch'] = rd, wr
wc = flat['SQ_WAVE_CYCLES']
flat['wave_cycle_split'] = {'parked(s_waitcnt/barrier)': flat['SQ_WAIT_ANY'] / wc, 'issue_stall': flat['SQ_WAIT_INST_ANY'] / wc, 'issuing': flat['SQ_ACTIVE_INST_ANY'] / wc}
flat['what'] = (f"tools/run_profiles.sh {RND}: tools/tools_pmc.sh {RND} spmm_tiled fetch write tcc sq (rocprofv3 --kernel-trace --pmc <group> -- "
                "python bench.py --steps 3 --warmup 1 --no-cpu --no-paths; one counter group per run); gfx950 correction: reads = 2 * FETCH_SIZE * 1024")
json.dump(flat, open(f'profiles/{RND}_tiled_pmc_summary.json', 'w'), indent=1)
t = json.load(open('profiles/traffic.json'))
t['round'] = NUM
k = t['kernels']['spmm_tiled']
k['round'] = NUM
k.update({'FETCH_SIZE_KB_per_launch': flat['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': flat['WRITE_SIZE'], 'traffic_bytes_per_launch': rd + wr,
          'TCC_HIT_sum': flat['TCC_HIT_sum'], 'TCC_MISS_sum': flat['TCC_MISS_sum'], 'TCC_REQ_sum': flat['TCC_REQ_sum'],
          'SQ_WAVE_CYCLES': wc, 'SQ_WAIT_ANY': flat['SQ_WAIT_ANY'], 'SQ_WAIT_INST_ANY': flat['SQ_WAIT_INST_ANY'], 'SQ_ACTIVE_INST_ANY': flat['SQ_ACTIVE_INST_ANY']})
k['rocprof_source'] = (f"profiles/{RND}_headline_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu "
                       f"--no-paths; the same run reports kernel_ms by HIP events in profiles/{RND}_headline_profiled_line.json)")
k['note'] = (f"round {NUM}: counters re-collected on this round's library (the fp32 executor's phase code is unchanged since late round 2; "
             "round 5 made its group index 32-bit and added the optional row map of balanced layouts)")
for r in csv.DictReader(open(f'profiles/{RND}_headline_kernel_stats.csv')):
    if 'spmm_tiled_kernel<0, 0, float>' in r['Name']:
        k['rocprof_kernel_avg_ms'], k['rocprof_kernel_calls'] = float(r['AverageNs']) / 1e6, int(r['Calls'])
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
d = json.loads(open(f'profiles/{RND}_bench_line.json').read())
p = json.loads(line)
print('bench line: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
print('profiled run: kernel_ms', p['roofline']['kernel_ms'], 'rocprof avg', k['rocprof_kernel_avg_ms'], 'calls', k['rocprof_kernel_calls'])
print('wave split', flat['wave_cycle_split'])
for extra in ('small_workloads.json',):
    if os.path.exists(f'{src}/{extra}'):
        shutil.copy(f'{src}/{extra}', f'profiles/{RND}_{extra}')
lines = [l for l in open(f'{src}/bench_stdout.txt') if l.startswith('{"paths"')]
if lines:
    open(f'profiles/{RND}_paths.json', 'w').write(json.dumps(json.loads(lines[-1])['paths'], indent=1))
t['note'] = f"round {NUM}: re-collected by tools/run_profiles.sh {RND} (see rocprof_source of each kernel)"
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
for a, v in d['roofline'].get('paths_ms', {}).items():
    print(' ', a, v, d['roofline']['paths_frac'].get(a))
