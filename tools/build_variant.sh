#!/usr/bin/env bash
# build_variant.sh <name> <defs...>: an ORL_BUILD_DEFS build of liborl_hip.so copied to variants/<name>.so (repo root; the
# directory is git-ignored and travels to the GPU box).  Restores nothing: run `python -m openrl_amd.csrc.build --force`
# afterwards to get the default build back in place.
set -e
name=$1; shift
mkdir -p variants
ORL_BUILD_DEFS="$*" python -m openrl_amd.csrc.build --force > /dev/null
cp openrl_amd/csrc/liborl_hip.so variants/$name.so
echo "variants/$name.so  <- $*"
