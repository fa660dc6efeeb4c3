for o in 6 4 3; do
MNAS_NT_MAX=$o MNAS_NO_SIDE=1 MNAS_BENCH_DETAIL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/nt_$o.txt 2> gpurun_out/nt_$o.err
MNAS_NT_MAX=$o python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/ntw_$o.txt 2>&1
done
