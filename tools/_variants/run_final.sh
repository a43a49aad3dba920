set -e
export TMPDIR=/tmp
ROOT=$(pwd)
TAG=${1:-r02_v7}
bash tools/profile_round.sh $TAG
timeout -k 10 300 python bench.py --scene ulica --width 1920 --height 1080 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_ulica1080_bench.json 2> gpurun_out/${TAG}_ulica_bench.err
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_ulica_trace -- python3 $ROOT/bench.py --scene ulica --width 1920 --height 1080 --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_ulica_trace.log 2>&1)
bash tools/pmc.sh raytracer_amd/librtgpu.so ${TAG}_ulica ulica 1920 1080
timeout -k 10 600 python tools/share_time.py 1 ulica 7680 4320 2>&1 | tail -1 | cut -c1-230 > gpurun_out/${TAG}_ulica8k.log
timeout -k 10 300 python tools/share_time.py 8 2>&1 | tail -2 | cut -c1-230 > gpurun_out/${TAG}_share8.log
