"""The host emulator's LDS-DMA model (tools/hipemu): a piece poisons its destination when it is issued, delivers its bytes when a vmcnt wait of the
issuing thread retires it (oldest first, as late as the count allows), s_barrier alone retires nothing, __syncthreads() and the end of the program
retire everything.  This is what turns an under-counted wait of a kernel into NaNs on the host (tools/hipemu/hipemu.h)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "hipemu.h"
#include <stdio.h>
#include <string.h>
static unsigned char lds[4][64];
static int fails = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)
int main() {
    unsigned char src[3][16];
    for (int i = 0; i < 3; ++i) memset(src[i], 0x10 + i, 16);
    hipemu::launch(dim3(1), dim3(2), 0, [&]() {
        const int t = threadIdx.x;
        unsigned char* d = lds[t];
        memset(d, 0x55, 64);                                   // the slot's previous tenant
        hipemu::dma_issue(d, src[0], 16);
        hipemu::dma_issue(d + 16, src[1], 16);
        hipemu::dma_issue(d + 32, nullptr, 4);                 // out-of-range piece: zeros
        EXPECT(d[0] == 0xFF && d[16] == 0xFF && d[32] == 0xFF && d[36] == 0x55);   // poisoned at issue, bytes outside the pieces untouched
        hipemu::barrier();                                     // s_barrier: no retirement
        EXPECT(d[0] == 0xFF);
        hipemu::dma_wait(2);                                   // at most two outstanding: the oldest has landed
        EXPECT(d[0] == 0x10 && d[15] == 0x10 && d[16] == 0xFF && d[32] == 0xFF);
        hipemu::dma_wait(2);                                   // nothing more to retire
        EXPECT(d[16] == 0xFF);
        if (t == 0) {
            __syncthreads();                                   // fence + barrier: everything of THIS thread retired
            EXPECT(d[16] == 0x11 && d[32] == 0x00 && d[35] == 0x00);
        } else {
            hipemu::dma_issue(d + 48, src[2], 16);             // still in flight when the thread ends: the end of the program retires it
            __syncthreads();
        }
    });
    EXPECT(lds[1][48] == 0x12 && lds[1][16] == 0x11);
    printf(fails ? "FAILED\n" : "OK\n");
    return fails != 0;
}
'''


@pytest.mark.skipif(not os.path.exists("/usr/bin/g++"), reason="needs g++")
def test_lds_dma_pieces_land_at_the_counted_wait(tmp_path):
    main = tmp_path / "dma_model.cpp"
    main.write_text(SRC)
    exe = tmp_path / "dma_model"
    hip = os.path.join(REPO, "tools", "hipemu")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", hip, str(main), os.path.join(hip, "hipemu.cpp"), "-o", str(exe)], check=True)
    env = {k: v for k, v in os.environ.items() if k != "HIPEMU_SYNC_DMA"}
    p = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout + p.stderr


RACY = r'''
#include "hipemu.h"
#include <stdio.h>
static int flag, seen;
int main() {
    flag = 0; seen = -1;
    hipemu::launch(dim3(1), dim3(128), 0, [&]() {          // wave 0 hands a value to wave 1 through memory WITHOUT a barrier
        if (threadIdx.x == 0) flag = 1;
        if (threadIdx.x == 64) seen = flag;
        __syncthreads();
    });
    printf("%d\n", seen);
    return 0;
}
'''


@pytest.mark.skipif(not os.path.exists("/usr/bin/g++"), reason="needs g++")
def test_wave_order_fuzzing_exposes_a_hand_over_without_a_barrier(tmp_path):
    """HIPEMU_ORDER=reverse / random runs the waves of a workgroup in another (legal) order between rendezvous points: a kernel whose result
    changes with it has an inter-wave race.  The emulator-based suite is order-independent (profiles/r05_emulator_race_screens.txt)."""
    main = tmp_path / "racy.cpp"
    main.write_text(RACY)
    exe = tmp_path / "racy"
    hip = os.path.join(REPO, "tools", "hipemu")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", hip, str(main), os.path.join(hip, "hipemu.cpp"), "-o", str(exe)], check=True)
    base = {k: v for k, v in os.environ.items() if k != "HIPEMU_ORDER"}
    out = {o: subprocess.run([str(exe)], capture_output=True, text=True, env={**base, **({"HIPEMU_ORDER": o} if o else {})}).stdout.strip()
           for o in ("", "reverse")}
    assert out[""] == "1" and out["reverse"] == "0", out
