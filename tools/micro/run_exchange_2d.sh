cd "$(dirname "$0")/../.."
echo "== 1-D pattern (the shipped decomposition), one stream, no arithmetic: vector of 64 KB pulled whole by every workgroup"
timeout -k 5 60 tools/micro/exchange_latency 2 128 32 16 0
timeout -k 5 60 tools/micro/exchange_latency 2 128 32 16 4
echo "== 2-D pattern"
for w in "0 0" "2 2"; do
for s in 2 4 8; do for ns in 1 2 4; do timeout -k 5 60 tools/micro/exchange_2d $s $ns $w || exit 1; done; done
done
