export MCRT_TUNING=1
for m in 2 6 14; do
  echo "mask $m"; MCRT_PACKET_BOUNCES=$m bash tools/configs.sh 2>&1 | cut -c1-120
done
