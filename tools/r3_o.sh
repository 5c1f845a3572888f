#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "apply_on_load" -x 2>&1 | tail -25
