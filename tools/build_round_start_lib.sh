# tools/lab/libexamg_r4a.so: libexamg with kernels_twostage.hip as it stood at the start of round 4's second session (commit 7e6a9d1), for the
# one-process A/B of tools/ab_libs.py against the current build (the .so is git-ignored and travels to the GPU box with the snapshot).
# The old file has no examg_jacobi3 / examg_rbgs_colours3 / examg_three_stage_eligible: stubs that return an error keep the ctypes bindings loadable.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
mkdir -p $T/exastencils_amd $T/include
cp -r $R/exastencils_amd/csrc $T/exastencils_amd/ && cp $R/include/examg.h $T/include/
git -C $R show 7e6a9d1:exastencils_amd/csrc/kernels_twostage.hip > $T/exastencils_amd/csrc/kernels_twostage.hip
cat >> $T/exastencils_amd/csrc/kernels_twostage.hip <<'EOF'

extern "C" int examg_jacobi3(const examg_layout_t *, const double *, double *, double *, const examg_layout_t *, const double *, const examg_stencil_t *, double,
                             const int32_t *, const int32_t *, examg_stream_t) { return 1; }
extern "C" int examg_rbgs_colours3(const examg_layout_t *, const double *, double *, const examg_layout_t *, const double *, const examg_stencil_t *, double, int,
                                   const int32_t *, const int32_t *, examg_stream_t) { return 1; }
extern "C" int examg_three_stage_eligible(const examg_layout_t *, const examg_layout_t *, const examg_stencil_t *, const int32_t *, const int32_t *) { return 0; }
EOF
(cd $T && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-function -o $R/tools/lab/libexamg_r4a.so exastencils_amd/csrc/*.hip exastencils_amd/csrc/*.cpp -ldl)
rm -rf $T; ls -la $R/tools/lab/libexamg_r4a.so
