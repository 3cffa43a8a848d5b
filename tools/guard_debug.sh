echo "== pytest first test"; timeout 300 python3 -m pytest tests/test_gpu_headline.py -x -q -k unsynchronised_profiled 2>&1 | grep -v "^  File\|amdgpu.ids" | head -30 | cut -c1-250
echo "== bench worker nosync"; timeout 120 python3 bench.py --worker --steps 3 --warmup 1 --no-prb --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids\|^  File" | head -20 | cut -c1-250
echo "== bench worker nosync again"; timeout 120 python3 bench.py --worker --steps 3 --warmup 1 --no-prb --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids\|^  File" | head -20 | cut -c1-250
