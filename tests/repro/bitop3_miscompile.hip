// Repro of the ROCm 7.2 hipcc (AMD LLVM 22.0.0git) gfx950 miscompile described in DESIGN.md §4:
//   hipcc --offload-arch=gfx950 -O3 bitop3_miscompile.hip && ./a.out   -> 'mode 1 mismatches 37 / 200000'
//   add  -Xclang -target-feature -Xclang -bitop3-insts               -> 0 mismatches (also 0 at -O0/-O1)
// The loop body's v_bitop3_b32 for the second borrow of dec_at gets truth table 0x84 instead of 0x04.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../backgammon-engine_amd/csrc/bg_board.h"
#include "../backgammon-engine_amd/csrc/bg_staged.h"
using namespace bg;
struct Case { uint32_t p[8]; uint32_t key; int pl, dA, dB; };
template<int MODE> __global__ void k(const Case* cs, int n, uint32_t* out)
{
    int i = blockIdx.x*blockDim.x+threadIdx.x; if (i>=n) return;
    Case c = cs[i];
    Side own, opp;
    for (int k2=0;k2<4;++k2){ own.b[k2] = c.pl ? c.p[4+k2] : c.p[k2]; opp.b[k2] = c.pl ? c.p[k2] : c.p[4+k2]; }
    int len = key_len(c.key);
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < len) apply_move(own, opp, c.pl, key_origin(c.key, q), (q & 1) ? c.dB : c.dA);
    } else {
#pragma unroll 1
        for (int q = 0; q < len; ++q)
            apply_move(own, opp, c.pl, key_origin(c.key, q), (q & 1) ? c.dB : c.dA);
    }
    for (int k2=0;k2<4;++k2){ out[i*8+k2] = c.pl ? opp.b[k2] : own.b[k2]; out[i*8+4+k2] = c.pl ? own.b[k2] : opp.b[k2]; }
}
// host reference: same arithmetic written independently on counts
static void host_apply(int cnt[2][26], int pl, int o, int d){
    int dest = pl ? o-d : o+d; if (dest<0) dest=0; if (dest>25) dest=25;
    cnt[pl][o]--; cnt[pl][dest]++;
    if (dest>=1 && dest<=24 && cnt[1-pl][dest]==1){ cnt[1-pl][dest]=0; cnt[1-pl][pl?0:25]++; }
}
int main(){
    srand(1); int N = 200000; std::vector<Case> cs(N); std::vector<uint32_t> exp(N*8);
    for (int i=0;i<N;++i){
        int cnt[2][26] = {{0}};
        for (int t=0;t<15;++t){ int pt; do { pt = 1 + rand()%24; } while (cnt[1][pt]); if (cnt[0][pt] < 15) cnt[0][pt]++; }
        for (int t=0;t<15;++t){ int pt; int tries=0; do { pt = 1 + rand()%24; } while (cnt[0][pt] && ++tries<100); if (!cnt[0][pt] && cnt[1][pt] < 15) cnt[1][pt]++; }
        Case c; for (int k2=0;k2<4;++k2){ uint32_t a=0,b=0; for (int q=0;q<26;++q){ a |= (uint32_t)((cnt[0][q]>>k2)&1)<<q; b |= (uint32_t)((cnt[1][q]>>k2)&1)<<q;} c.p[k2]=a; c.p[4+k2]=b; }
        c.pl = rand()&1; c.dA = 1+rand()%6; c.dB = 1+rand()%6; int len = 1 + rand()%3; uint32_t key = 0;
        for (int q=0;q<len;++q){ int o; int tries=0; do { o = 1+rand()%24; } while (cnt[c.pl][o]==0 && ++tries<1000); if (cnt[c.pl][o]==0) { len=q; break; }
            key = key_child(key, o); host_apply(cnt, c.pl, o, (q&1)? c.dB : c.dA); }
        c.key = key; cs[i]=c;
        for (int k2=0;k2<4;++k2){ uint32_t a=0,b=0; for (int q=0;q<26;++q){ a |= (uint32_t)((cnt[0][q]>>k2)&1)<<q; b |= (uint32_t)((cnt[1][q]>>k2)&1)<<q;} exp[i*8+k2]=a; exp[i*8+4+k2]=b; }
    }
    Case* d; uint32_t* o; hipMalloc(&d, N*sizeof(Case)); hipMalloc(&o, N*32); hipMemcpy(d, cs.data(), N*sizeof(Case), hipMemcpyHostToDevice);
    std::vector<uint32_t> got(N*8);
    for (int mode=0; mode<2; ++mode){
        if (mode==0) hipLaunchKernelGGL(k<0>, dim3((N+255)/256), dim3(256), 0, 0, d, N, o); else hipLaunchKernelGGL(k<1>, dim3((N+255)/256), dim3(256), 0, 0, d, N, o);
        hipMemcpy(got.data(), o, N*32, hipMemcpyDeviceToHost);
        int bad=0; for (int i=0;i<N;++i){ bool b=false; for (int q=0;q<8;++q) if (got[i*8+q]!=exp[i*8+q]) b=true; if (b && bad++<3) { printf("mode %d case %d key %08x pl %d dA %d dB %d\n in : ", mode, i, cs[i].key, cs[i].pl, cs[i].dA, cs[i].dB); for (int q=0;q<8;++q) printf("%08x ", cs[i].p[q]); printf("\n exp: "); for (int q=0;q<8;++q) printf("%08x ", exp[i*8+q]); printf("\n got: "); for (int q=0;q<8;++q) printf("%08x ", got[i*8+q]); printf("\n"); } }
        printf("mode %d mismatches %d / %d\n", mode, bad, N);
    }
}
