#!/bin/bash
# tools/with_variant.sh <name> <command...>: run the command with smartdenovo_amd/variants/libwtzmo_hip_<name>.so in the product library's place (GPU box only: the copy is scratch)
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
L=$R/smartdenovo_amd/libwtzmo_hip.so
cp $L /tmp/libwtzmo_hip.orig.$$ && cp $R/smartdenovo_amd/variants/libwtzmo_hip_$N.so $L
"$@"; rc=$?
cp /tmp/libwtzmo_hip.orig.$$ $L; rm -f /tmp/libwtzmo_hip.orig.$$
exit $rc
