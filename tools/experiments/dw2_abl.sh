for a in 0 1 2 4 6 7; do echo "ablate $a"; GTG_DW2_ABLATE=$a timeout -k 10 120 python tools/step_prof.py --tags dw2 --steps 20 | grep dw2_sparse || exit 1; done
