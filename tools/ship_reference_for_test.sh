#!/bin/bash
# Copies the few reference files the reference's sampler stack imports into a git-ignored, TEMPORARY directory of the repo so that one gpurun call
# can run tests/test_reference_hosted_gpu.py on the GPU box (which has no /root/reference):   tools/ship_reference_for_test.sh stage | clean
# Never committed (.gitignore: _ref_testdata/); removed again right after the call.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
dst="$root/_ref_testdata"
case "$1" in
  stage)
    src=${VISTA_REFERENCE:-/root/reference}
    mkdir -p "$dst/vwm/modules/diffusionmodules"
    cp "$src/vwm/util.py" "$dst/vwm/"
    # (round 5: + the network itself -- video_model / openaimodel / attention / video_attention -- for the reference's own fp16-autocast error)
    for f in denoiser.py denoiser_scaling.py denoiser_weighting.py discretizer.py guiders.py sampling.py sampling_utils.py wrappers.py util.py video_model.py openaimodel.py; do
      [ -f "$src/vwm/modules/diffusionmodules/$f" ] && cp "$src/vwm/modules/diffusionmodules/$f" "$dst/vwm/modules/diffusionmodules/"
    done
    cp "$src/vwm/modules/attention.py" "$src/vwm/modules/video_attention.py" "$dst/vwm/modules/"
    echo "staged under $dst: run with VISTA_REFERENCE=$dst" ;;
  clean) rm -rf "$dst" ;;
  *) echo "usage: $0 stage|clean"; exit 2 ;;
esac
