# same-box A/B of two builds of the library: tools/exp_ab_lib.sh <a.so> <b.so> -- <python script and args>
A=$1; B=$2; shift 3
for rep in 1 2 3; do
  for L in $A $B; do echo "== $L"; RINGHIP_LIB=$PWD/$L python3 "$@" | tail -${TAILN:-1}; done
done
