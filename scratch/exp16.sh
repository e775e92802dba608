for r in 0 1 2; do echo "radius $r"; BHRAY_TEMPORAL_RADIUS=$r python scratch/exp15.py 2>&1 | grep orbit; done
timeout 600 python -m pytest tests/test_gpu_temporal.py -x -q 2>&1 | tail -3
