#!/bin/bash
# builds a variant of the library for an A/B run on one box: bash profiles/ab_build.sh <name> [hipcc flags...] -> sedef_amd/lib/ab/<name>.so
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function "$@" -o sedef_amd/lib/ab/$name.so sedef_amd/csrc/sdf_unity.hip 2>&1 | grep -E "error" ; echo "$name built"
