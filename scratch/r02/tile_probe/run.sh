#!/bin/bash
cd "$(dirname "$0")"
for p in base late nobar nogl nosw noglsw nofrag mfmaonly; do echo "== probe_$p"; timeout -k 10 120 ./probe_$p || exit 1; done
