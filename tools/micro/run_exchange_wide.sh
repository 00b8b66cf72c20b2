#!/bin/bash
# sweep of tools/micro/exchange_wide: streams, emulated work, scope bits, request position, one or two batch groups
cd "$(dirname "$0")/../.."
X="timeout -k 5 60 tools/micro/exchange_wide"
echo "== no arithmetic at all"
for ns in 1 2 4; do $X $ns 0 0 16 16 0 1 || exit 1; done
echo "== emulated work (mfma 2 sleeps ~0.42 us, cell 2 sleeps ~0.42 us), shipped scopes (sc1 loads, sc1 stores)"
for ns in 1 2 4; do $X $ns 2 2 16 16 0 1 || exit 1; done
for ns in 2 4; do $X $ns 2 2 16 16 1 1 || exit 1; done
echo "== two batch groups (256 workgroups)"
for ns in 2 4; do $X $ns 2 2 16 16 0 2 || exit 1; done
echo "== scope bits: loads sc0 sc1 (17), sc0 (1), nt sc1 (18); stores sc0 sc1 (17), nt sc1 (18)"
for la in 17 1 18; do $X 2 2 2 $la 16 0 1 || exit 1; done
for sa in 17 18; do $X 2 2 2 16 $sa 0 1 || exit 1; done
echo "== longer emulated work (mfma 3, cell 3)"
for ns in 2 4; do $X $ns 3 3 16 16 0 1 || exit 1; $X $ns 3 3 16 16 1 1 || exit 1; done
