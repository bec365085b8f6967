for cfg in "22016,4096 16 6" "22016,4096 16 3" "22016,4096 8 3" "12288,4096 16 3" "12288,4096 8 2" "11008,4096 16 3" "4096,11008 16 1" "4096,11008 8 1"; do
  set -- $cfg
  for d in 2 4; do
    echo "== shape $1 waves $2 rpt $3 depth $d"
    tools/mb_short.sh --only $1 --waves $2 --rpt $3 --depth $d || exit 1
  done
done
