#!/bin/bash
# GPU box: the memory-only probes behind DESIGN.md sections 3-5 (build them first: hipcc --offload-arch=gfx950 -O3 -std=c++17
# tools/dev/<name>.hip -o tools/variants/<name>)
for p in rows_probe3 rows_probe2 rows_probe_f64 mfma_f64_probe; do echo "== tools/dev/$p.hip"; timeout -k 5 120 tools/variants/$p; done
