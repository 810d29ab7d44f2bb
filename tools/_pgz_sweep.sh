cd /tmp/pgzb 2>/dev/null || { bash /root/repo/tools/pgz_bench.sh 2000000 > /dev/null 2>&1; cd /tmp/pgzb; }
for t in 16 32; do for c in 131072 262144 524288 1048576 2097152; do for p in 1 2 4; do
  echo "threads $t chunk $c cpt $p: $(FFQ_PGZ_CHUNK=$c FFQ_PGZ_CPT=$p ./pgz_main a6.gz $t | tr '\n' ' ' | cut -c1-250)"
done; done; done
