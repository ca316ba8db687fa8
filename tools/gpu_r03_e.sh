cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_serving.py -q -m gpu -s > $O/r03_pytest_serving.log 2>&1; tail -40 $O/r03_pytest_serving.log
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_serving.py > $O/r03_pytest_rest.log 2>&1; tail -5 $O/r03_pytest_rest.log
