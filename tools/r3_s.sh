#!/bin/bash
python tools/ln_linear_time.py 2>&1 | grep -v amdgpu.ids
