# usage: bash profiles/microbench/resources.sh [n_mass ...]  — registers / spills / scratch / LDS of every kernel AND phase function of the library
# (hipcc -Rpass-analysis=kernel-resource-usage, device code only); no argument: the C ABI unit (cartpole / linear-system / library kernels)
cd "$(dirname "$0")/../../mpc4rl_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -Rpass-analysis=kernel-resource-usage --cuda-device-only"
one() {
  /opt/rocm/bin/hipcc $FLAGS "$@" -o /dev/null 2>&1 | python3 -c "
import re, sys
cur = {}
def flush():
    if cur.get('name'):
        n = cur['name']
        n = re.sub(r'^_ZN5mpcrl', '', n)
        print('%-110s VGPR %3s AGPR %3s VGPR-spill %4s SGPR-spill %4s scratch %5s occ %s LDS %6s' % (n[:110], cur.get('VGPRs','?'), cur.get('AGPRs','?'), cur.get('VGPRs Spill','?'), cur.get('SGPRs Spill','?'), cur.get('ScratchSize [bytes/lane]','?'), cur.get('Occupancy [waves/SIMD]','?'), cur.get('LDS Size [bytes/block]','?')))
for line in sys.stdin:
    m = re.search(r'remark: +(.*?): (.*?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ('Function Name', 'Name'):
        flush(); cur = {'name': v}
    else:
        cur[k] = v
flush()
"
}
if [ $# -eq 0 ]; then one -c mpcrl_api.hip; else for n in "$@"; do one -DMPCRL_CHAIN_NMASS=$n -c chain_inst.hip; done; fi
