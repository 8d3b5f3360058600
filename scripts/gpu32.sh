export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "usage_scenarios" 2>&1 | grep -v Warning | tail -15
