#!/bin/bash
# Timing-only builds of the generated feed-forward forward loop (tools/gen_ffn_fwd.py --abl n: 1 no MFMA, 2 no ReLU / mask VALU, 4 no fragment
# reads, 8 no LDS-DMA, 16 no hid / mask stores; results garbage).  Built HERE (hipcc cross-compiles), run on the GPU box:
#   tools/abl_ffn2.sh build 1 4 8 ...     ->  end-to-end_asr_pytorch_amd/csrc/build/abl/libasr_ffn2_<n>.so
#   tools/abl_ffn2.sh run 0 1 4 8 ...     (0 = the product library)
CS=end-to-end_asr_pytorch_amd/csrc
mode=$1; shift
mkdir -p $CS/build/abl
for spec in "$@"; do
  IFS=, read abl pol <<< "$spec"        # "119,sc0+nt": ablation bits, cache policy of the LDS-DMA loads
  tag=$abl${pol:+_$pol}
  so=$CS/build/abl/libasr_ffn2_$tag.so
  if [ "$mode" = build ]; then
    python tools/gen_ffn_fwd.py --abl $abl ${pol:+--policy $pol} --out $CS/build/abl/ffn_fwd2_$tag.inc 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFFN2_FWD_INC="\"build/abl/ffn_fwd2_$tag.inc\"" -c $CS/ffn2.hip -o $CS/build/abl/ffn2_$tag.o 2>$CS/build/abl/ffn2_$tag.err || { echo "$spec: compile failed"; grep -m3 error $CS/build/abl/ffn2_$tag.err; continue; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $so $CS/build/abl/ffn2_$tag.o $(ls $CS/build/*.o | grep -v "/ffn2.hip.o")
    echo "built $so"
  else
    if [ $(( abl & 64 )) -ne 0 ]; then r=$(ASR_AMD_LIB=$PWD/$so timeout 200 python3 tools/check_ffn2.py --stamps 2>&1 | grep "^loop" | tr '\n' ';')
    elif [ "$abl" = 0 ]; then r=$(timeout 200 python3 tools/check_ffn2.py --time-only 2>&1 | grep "^ffn_fwd" | tr '\n' ';')
    else r=$(ASR_AMD_LIB=$PWD/$so timeout 200 python3 tools/check_ffn2.py --time-only 2>&1 | grep "^ffn_fwd" | tr '\n' ';'); fi
    echo "FFN2_ABL=$spec  $r"
  fi
done
