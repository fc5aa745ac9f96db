#!/bin/bash
# Builds ab/lib<NAME>.so from the current sources with extra compiler flags, for scripts/ab_libs.py / scripts/ab.sh:
#   scripts/build_variant.sh NAME "-DBRT_SHARED_RCP=0"      (objects under bevyray_amd/csrc/build_<NAME>/, removed afterwards)
set -e
name="$1"; flags="$2"
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/ab"
make -s -C "$root/bevyray_amd/csrc" -j8 EXTRA="$flags" BUILD="build_$name" OUT="$root/ab/lib$name.so"
rm -rf "$root/bevyray_amd/csrc/build_$name"
ls -la "$root/ab/lib$name.so"
