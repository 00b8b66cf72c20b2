cd "$(dirname "$0")/../.."
echo "== shipped layout: 2 groups (directions) x 128 workgroups, 512 B published per workgroup and stream-step, 64 KB pulled; work 0 / 4 x 0.21 us"
for w in 0 4; do timeout -k 5 60 tools/micro/exchange_latency 2 128 32 16 $w; done
echo "== fat workgroups, two batches side by side: 4 groups (2 batches x 2 directions) x 64 workgroups, 1 KB published, 64 KB pulled; work 0 / 6"
for w in 0 6; do timeout -k 5 60 tools/micro/exchange_latency 4 64 64 16 $w; done
echo "== fat workgroups, one batch on half of the CUs: 2 groups x 64 workgroups; work 0 / 6"
for w in 0 6; do timeout -k 5 60 tools/micro/exchange_latency 2 64 64 16 $w; done
