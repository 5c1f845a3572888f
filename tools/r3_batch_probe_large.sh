#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_conv.py -q -m gpu 2>&1 | tail -8
