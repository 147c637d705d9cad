#!/bin/bash
# what the PMC passes of round 5 profile in ONE rocprofv3 invocation: four copy forms and two forms of the product kernel
cd "$GRAFT_REPO_ROOT"
tools/window_lab 2 'flat copy|seg 128&static far&copy|seg 128&dynamic per workgroup&copy|seg  64&static far&copy|seg  64&dynamic per PAIR &copy'
tools/p64v_bench 1 'shipped (3,3) spreads|pair tickets, ONE counter, uncached'
