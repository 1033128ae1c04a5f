for v in "$@"; do
  if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep --mode exact > gpurun_out/abx_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/abx_$v.log; continue; }
  tail -1 gpurun_out/abx_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact $v', round(d['value']), d['roofline']['avg_launch_ms'])"
done
