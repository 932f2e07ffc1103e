mkdir -p gpurun_out/c
for cfg in "1 3" "1 4" "1 6" "2 1" "2 2" "2 3" "3 1" "3 2" "4 1" "4 2" "6 1" "6 2"; do set -- $cfg
  python bench.py --batch $1 --streams $2 --steps 40 --no-cpu-baseline --no-replay 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('batch',c['msms_per_step'],'streams',c['streams'],'value %.1f M pairs/s'%(d['value']/1e6),'ms/step %.3f'%d['ms_per_step'],'per MSM %.3f'%(d['ms_per_step']/c['msms_per_step']),'check',c['check'],'issue %.3f'%d['roofline']['issue']['frac'])"
done > gpurun_out/c/batch_sweep.txt 2>&1
cat gpurun_out/c/batch_sweep.txt
