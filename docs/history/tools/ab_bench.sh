# A/B on ONE box (boxes differ by several %): bench of the committed baseline copy (_ab/base, made by tools/ab_prepare.sh) vs the working tree, alternating.
for i in 1 2; do
  (cd _ab/base && python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('base', l['value'], l['ms_per_step'])")
  python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('new ', l['value'], l['ms_per_step'])"
done
