mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
amd-smi metric --help > $O/amdsmi_metric_help.txt 2>&1
amd-smi monitor --help > $O/amdsmi_monitor_help.txt 2>&1
amd-smi metric -g 0 --json > $O/amdsmi_metric_idle.json 2>&1
amd-smi static -g 0 --json > $O/amdsmi_static.json 2>&1
ls -la /sys/class/drm/ > $O/sysfs.txt 2>&1
for c in /sys/class/drm/card*/device; do echo $c; ls $c | tr '\n' ' '; echo; ls $c/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done >> $O/sysfs.txt 2>&1
for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo $f $(cat $f 2>&1); done >> $O/sysfs.txt
python - >> $O/sysfs.txt 2>&1 <<'P'
import glob,time
fs=glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input')+glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average')
print(fs)
for f in fs:
    t=time.time(); n=0; vals=set()
    while time.time()-t<0.5:
        vals.add(open(f).read().strip()); n+=1
    print(f, n, 'reads in 0.5 s', len(vals), 'distinct')
gm=glob.glob('/sys/class/drm/card*/device/gpu_metrics')
print(gm)
for f in gm:
    b=open(f,'rb').read(); print(f,len(b),b[:4].hex())
P
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "output_nc3" 2>&1 | tail -5 > $O/nc3_test.log
APAMD_PRECISION=bf16 python tools/train_bench.py 16 5 > $O/train_bf16.log 2>&1
