cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
for g in 16 32; do ORC_FAST_GROUP=$g python bench.py --steps 3 --warmup 1 --no-e2e --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['cpu_baseline']; print('group $g', c['value'], c['probes_per_s_per_core_M'], c['plain_form_Mreads_s'], c['parity_with_gpu_on_sample'])"; done
grep -i hugepages /proc/meminfo
