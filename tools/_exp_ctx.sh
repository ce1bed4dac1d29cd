echo "--- F: cpp_host + golden"
python -m pytest tests/test_cpp_host.py tests/test_golden_frames.py -q -m gpu -x 2>&1 | tail -3
echo "--- G: golden + multiproc (collection imports torch first)"
python -m pytest tests/test_golden_frames.py tests/test_gpu_multiproc.py -q -m gpu -x 2>&1 | tail -3
echo "--- H: cpp_host + golden + multiproc"
python -m pytest tests/test_cpp_host.py tests/test_golden_frames.py tests/test_gpu_multiproc.py -q -m gpu -x 2>&1 | tail -3
echo "--- I: everything but multiproc"
python -m pytest tests -q -m gpu -x --ignore tests/test_gpu_multiproc.py 2>&1 | tail -3
