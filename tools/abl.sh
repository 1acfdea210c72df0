for ab in 0 1 2 4 8 3 7 15; do
HE355_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$ab -- python3 bench.py --steps 2 --warmup 0 --profile-mode --chunk 128 > /dev/null 2>&1
f=$(find gpurun_out/abl_$ab -name "*kernel_stats.csv" | head -1)
echo "ablate=$ab $(grep 'k_k3<he355::ArF64>' $f | cut -d, -f3,4) u64: $(grep 'k_k3<he355::ArU64>' $f | cut -d, -f4)"
done
