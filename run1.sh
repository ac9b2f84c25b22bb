for cp in 1 0; do
PORESEQ_DEBUG_CPAIR=$cp timeout 900 python bench.py --no-cpu --no-extras --steps 2 --warmup 1 --regions-per-gpu 128 --batches-in-flight 8 2>/dev/null | tail -1 | cut -c1-100
PORESEQ_DEBUG_CPAIR=$cp timeout 900 python bench.py --no-cpu --no-extras --steps 2 --warmup 1 --regions-per-gpu 64 --batches-in-flight 4 2>/dev/null | tail -1 | cut -c1-100
done
