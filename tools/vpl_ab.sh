run() { tag=$1; shift
  python bench.py --no-planesweep --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', r['value'], r['ms_per_step'], 'rows', r['config']['tile_rows'], 'vpl', r['config']['views_per_launch'])"
}
for vpl in 0 16; do
run "exact_k7 vpl=$vpl" --mode exact --steps 3 --views-per-launch $vpl
run "k9 vpl=$vpl" --patch 9 --steps 3 --views-per-launch $vpl
run "k11 vpl=$vpl" --patch 11 --steps 3 --views-per-launch $vpl
run "k5 vpl=$vpl" --patch 5 --steps 3 --views-per-launch $vpl
run "k3 vpl=$vpl" --patch 3 --steps 3 --views-per-launch $vpl
run "1440p vpl=$vpl" --height 1440 --width 2560 --steps 2 --views-per-launch $vpl
done
run "4k8 vpl=0" --views-per-gpu 8 --height 2160 --width 3840 --steps 2
run "4k8 vpl=8" --views-per-gpu 8 --height 2160 --width 3840 --steps 2 --views-per-launch 8
