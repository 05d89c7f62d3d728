#!/bin/bash
# Builds a VARIANT of the library from the working tree with temporary source edits, next to the shipped one (CPU container):
#   tools/ab_variant.sh <name> '<file>:<sed expression>' ...     ->  melonix_amd/lib/variants/<name>.so
# then on the GPU box:  MX_AB_LIB=melonix_amd/lib/variants/<name>.so python tools/stft_sizes.py 32768x375 ...
# The product sources are not touched: A/B experiments leave no switches behind.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/melonix_amd" "$root/melonix_amd/lib/variants"
cp -r "$root/include" "$tmp/"
cp "$root"/melonix_amd/*.py "$tmp/melonix_amd/"
cp -r "$root/melonix_amd/csrc" "$tmp/melonix_amd/"
# reuse the objects of the units the edits do not touch
mkdir -p "$tmp/melonix_amd/build"; cp "$root"/melonix_amd/build/*.o "$tmp/melonix_amd/build/" 2>/dev/null || true
for e in "$@"; do
  f="${e%%:*}"; x="${e#*:}"
  before=$(md5sum "$tmp/melonix_amd/csrc/$f")
  sed -i -E "$x" "$tmp/melonix_amd/csrc/$f"
  [ "$before" != "$(md5sum "$tmp/melonix_amd/csrc/$f")" ] || { echo "edit had no effect: $e"; exit 1; }
done
(cd "$tmp" && python -c "import melonix_amd.build as b; b.build()")
cp "$tmp/melonix_amd/lib/libmelonix_amd.so" "$root/melonix_amd/lib/variants/$name.so"
rm -rf "$tmp"
echo "built variant $name -> melonix_amd/lib/variants/$name.so"
