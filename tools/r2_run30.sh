#!/bin/bash
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12
