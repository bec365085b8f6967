for cfg in "4096,4096 8 1 2" "4096,4096 8 1 4" "4096,4096 16 1 2" "12288,4096 8 1 2" "12288,4096 8 1 4" "22016,4096 8 2 2" "22016,4096 8 2 4" "22016,4096 8 3 4" "4096,11008 16 1 2" "4096,11008 16 1 4"; do
  set -- $cfg
  echo "== shape $1 waves $2 rpt $3 depth $4"
  tools/attic/mb_short.sh --only $1 --waves $2 --rpt $3 --depth $4 || exit 1
done
