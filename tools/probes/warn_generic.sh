export VVHIP_WARN_GENERIC=1
for c in C1 C2 C3 C4 C5; do echo "== $c"; python bench.py --config $c --steps 200 --warmup 20 --large-n none --no-cpu-baseline 2>&1 | grep "vvhip: kernel" | sort -u; done
echo "== C3 hbonds"; python bench.py --config C3 --hbonds --steps 200 --warmup 20 --large-n none --no-cpu-baseline 2>&1 | grep "vvhip: kernel" | sort -u
echo "== C3x8"; python bench.py --config C3x8 --steps 100 --warmup 20 --large-n none --no-cpu-baseline 2>&1 | grep "vvhip: kernel" | sort -u
echo "== C3x8 hbonds"; python bench.py --config C3x8 --hbonds --steps 100 --warmup 20 --large-n none --no-cpu-baseline 2>&1 | grep "vvhip: kernel" | sort -u
echo "== tests"; python -m pytest tests/test_gpu_steps.py -m gpu -q -s 2>&1 | grep "vvhip: kernel" | sort | uniq -c | sort -rn | head -20
