set -e
mkdir -p gpurun_out/r02_base
python bench.py --inflight 1 --no-cpu-baseline --steps 10 > gpurun_out/r02_base/english_serial.json 2> gpurun_out/r02_base/english_serial.err
python bench.py --inflight 1 --no-cpu-baseline --steps 10 --workload mixed --docs-per-gpu 25000 > gpurun_out/r02_base/mixed_serial.json 2> gpurun_out/r02_base/mixed_serial.err
cat gpurun_out/r02_base/english_serial.json gpurun_out/r02_base/mixed_serial.json
