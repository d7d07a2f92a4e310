# interleaved A/B of one bench flag at 30 / 60 sequences:  bash tools/exp/ab_flag_b30.sh "<flag>"
flag="$1"
for r in 1 2 3; do
  for b in 30 60; do
    for f in "" "$flag"; do
      python bench.py --batch $b --steps 30 --no-padded --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b$b', '[$f]', d['value'], d['ms_per_step'])"
    done
  done
done
