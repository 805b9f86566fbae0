#!/bin/bash
# Is the 32-bit build's device code what it was at <commit>?  Compiles both kernel files of that commit and of the working tree to
# gfx950 assembly (no GPU needed) and compares them with the compile-unit id lines taken out.  usage: isa_identity.sh [<commit>]
# (default 82e93df: the tree of round 5's last hg19-scale measurements).  profiles/r5/isa_identity_32bit.txt is its record.
set -e
cd "$(dirname "$0")/../.."
base=${1:-82e93df}
tmp=$(mktemp -d)
git worktree add -q --detach "$tmp/old" "$base"
for f in bt_kernels bt_best_kernels; do
	hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S "$tmp/old/bowtie_amd/csrc/$f.hip" -o "$tmp/$f.old.s" 2>/dev/null
	hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S "bowtie_amd/csrc/$f.hip" -o "$tmp/$f.new.s" 2>/dev/null
	n=$(diff <(grep -v __hip_cuid "$tmp/$f.old.s") <(grep -v __hip_cuid "$tmp/$f.new.s") | wc -l)
	echo "$f.hip: $(wc -l < "$tmp/$f.new.s") lines, $n differ from $base"
done
git worktree remove --force "$tmp/old"
rm -rf "$tmp"
