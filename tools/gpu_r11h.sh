#!/bin/bash
# visit: launch timelines of one build on the final code — device entry and host entry (direct upload)
tools/gpu_timeline.sh r11h_configC
TIMELINE_MARKER=MaskTableFunctor tools/gpu_timeline.sh r11h_host_entry_configC --host-entry
grep -c . gpurun_out/r11h_host_entry_configC_timeline.txt; grep "insert_wave\|MaskTable\|copyBuffer" gpurun_out/r11h_host_entry_configC_timeline.txt | head -30
