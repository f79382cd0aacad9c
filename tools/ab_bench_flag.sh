# Whole-step A/B of one engine flag through bench.py (same box, alternating): bash tools/ab_bench_flag.sh <flag> <v0> <v1> [extra bench args]
flag=$1; a=$2; b=$3; shift 3
for v in $a $b $a $b; do
  python bench.py --engine-flag $flag=$v --no-extras --no-cpu-baseline --no-roofline --steps 40 --warmup 6 "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
  python -c "import json; d=json.load(open('/tmp/ab_line.json')); print('$flag', $v, d['value'], d['ms_per_step'])"
done
