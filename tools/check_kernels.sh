#!/bin/bash
# List every gfx950 kernel of libhtrvt_hip.so that uses scratch memory or spills VGPRs (none should: a spilling GEMM
# kernel runs several times slower).  Usage: tools/check_kernels.sh [path/to/lib.so]
set -e
LIB=${1:-$(dirname "$0")/../htr-vt_amd/lib/libhtrvt_hip.so}
TMP=$(mktemp -d)
cp "$LIB" "$TMP/lib.so"
( cd "$TMP" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1 )
bad=0
for f in "$TMP"/lib.so.*gfx950; do
  out=$(/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | grep -E "^\s+\.name:|\.private_segment_fixed_size:|\.vgpr_spill_count:|\.vgpr_count:" | paste - - - - |
        awk '{ n=""; p=0; s=0; v=0; for (i=1;i<=NF;i++) { if ($i==".name:") n=$(i+1); if ($i==".private_segment_fixed_size:") p=$(i+1); if ($i==".vgpr_spill_count:") s=$(i+1); if ($i==".vgpr_count:") v=$(i+1) } if (p>0 || s>0) print n, "scratch", p, "vgpr_spills", s, "vgprs", v }')
  if [ -n "$out" ]; then echo "$out"; bad=1; fi
done
rm -rf "$TMP"
[ $bad -eq 0 ] && echo "no kernel uses scratch or spills"
exit 0
