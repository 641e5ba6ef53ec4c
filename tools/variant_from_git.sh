#!/bin/bash
# build the library as of a git revision into hvqm4_amd/abl/libhvq_<name>.so (same-box A/B against the working tree)
# usage: tools/variant_from_git.sh <name> <rev>
set -e
name=$1; rev=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
wt=/tmp/wt_$name
rm -rf $wt; git -C $root worktree add -f $wt $rev > /dev/null 2>&1
make -C $wt/hvqm4_amd/csrc > /tmp/wt_$name.log 2>&1
mkdir -p $root/hvqm4_amd/abl && cp $wt/hvqm4_amd/libhvqm4_amd.so $root/hvqm4_amd/abl/libhvq_$name.so
git -C $root worktree remove --force $wt; git -C $root worktree prune
echo "built hvqm4_amd/abl/libhvq_$name.so from $rev"
