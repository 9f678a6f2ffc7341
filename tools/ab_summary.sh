#!/bin/bash
# ab_summary.sh DIR: best value per variant of every ab_variants.sh listing in DIR
for f in "$1"/c*.txt; do echo "== $(basename $f .txt)"; awk '{n=split($0,a," "); v=a[n-1]; k=$0; sub(/: [0-9.]+ Mpaths\/s$/,"",k); if(!(k in m)||v>m[k])m[k]=v} END{for(k in m)print m[k], k}' $f | sort -rn; done
