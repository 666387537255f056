#!/bin/bash
# tools/gr.sh '<command>' [timeout]: gpurun with the scratch directory created first (gpurun_out/ does not travel)
exec timeout $(( ${2:-900} + 900 )) /usr/local/graft/bin/gpurun --timeout ${2:-900} -- "mkdir -p gpurun_out/r02 && $1"
