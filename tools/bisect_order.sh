# which earlier test file makes a later test fail?  bash tools/bisect_order.sh <test id> <file> <file> ...
cd $GRAFT_REPO_ROOT
t=$1; shift
for f in "$@"; do
  r=$(python -m pytest $f "$t" -m gpu -q -p no:cacheprovider 2>&1 | tail -1)
  echo "$f: $r"
done
