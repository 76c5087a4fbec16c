#!/bin/sh
# Build the host-emulation of libwtzmo_hip (every kernel run as a host loop) and the host driver linked to it.
# TEST INFRASTRUCTURE ONLY: lets the host driver / batching / commit logic be exercised without a GPU.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
# nothing to do when every product is newer than every source (several test modules call this script)
if [ -x "$HERE/wtzmo_emul" ] && [ -x "$HERE/wtgbo_emul" ] && [ -x "$HERE/wtext_emul" ] && [ -f "$HERE/libwtz_emul.so" ] && [ -f "$HERE/libwtzmo_host_emul.so" ]; then
  NEWER=$(find "$ROOT/smartdenovo_amd/csrc" "$ROOT/include" "$HERE/build_emul.sh" -type f -newer "$HERE/wtext_emul" | head -1)
  OLDEST_OK=1; for f in "$HERE/wtzmo_emul" "$HERE/wtgbo_emul" "$HERE/libwtz_emul.so" "$HERE/libwtzmo_host_emul.so"; do [ -n "$(find "$ROOT/smartdenovo_amd/csrc" "$ROOT/include" -type f -newer "$f" | head -1)" ] && OLDEST_OK=0; done
  if [ -z "$NEWER" ] && [ "$OLDEST_OK" = "1" ]; then exit 0; fi
fi
g++ -std=c++17 -O2 -g -DWTZ_EMUL -ffp-contract=off -Wall -Wno-unused-function -Wno-unknown-pragmas -I"$ROOT/include" -shared -fPIC \
    -o "$HERE/libwtz_emul.so" "$ROOT/smartdenovo_amd/csrc/wtz_lib.cpp"
gcc -std=gnu11 -O2 -g -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -Wno-sign-compare -I"$ROOT/include" \
    -o "$HERE/wtzmo_emul" "$ROOT/smartdenovo_amd/csrc/host/wtzmo_main.c" -L"$HERE" -lwtz_emul -Wl,-rpath,'$ORIGIN' -lstdc++ -lm -lpthread
# the same host driver as a shared object (wtzmo_main + wtzmo_set_dist + step hook) on the emulated device layer: rank tests load it with ctypes
gcc -std=gnu11 -O2 -g -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -Wno-sign-compare -DWTZ_AS_LIB -shared -fPIC -I"$ROOT/include" \
    -o "$HERE/libwtzmo_host_emul.so" "$ROOT/smartdenovo_amd/csrc/host/wtzmo_main.c" -L"$HERE" -lwtz_emul -Wl,-rpath,'$ORIGIN' -lstdc++ -lm -lpthread
# the drop-in wtgbo on the emulated device layer
gcc -std=gnu11 -O2 -g -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -Wno-sign-compare -I"$ROOT/include" \
    -o "$HERE/wtgbo_emul" "$ROOT/smartdenovo_amd/csrc/host/wtgbo_main.c" -L"$HERE" -lwtz_emul -Wl,-rpath,'$ORIGIN' -lstdc++ -lm -lpthread
# the drop-in wtext (f2) on the emulated device layer
gcc -std=gnu11 -O2 -g -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -Wno-sign-compare -Wno-unused-function -I"$ROOT/include" \
    -o "$HERE/wtext_emul" "$ROOT/smartdenovo_amd/csrc/host/wtext_main.c" -L"$HERE" -lwtz_emul -Wl,-rpath,'$ORIGIN' -lstdc++ -lm -lpthread
