// Probe (one GPU, two processes): can a kernel of process A store straight into device memory of process B through an IPC
// mapping, signal through a flag in page-locked shared host memory, and can a kernel of B wait on that flag?
// build: hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip      run: ./ipc_probe
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_fill(double *p, size_t n, double v) { size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (k < n) p[k] = v + (double)k; }
__global__ void k_signal(unsigned *flag, unsigned seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_wait(unsigned *flag, unsigned seq, unsigned *err) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0; it < (1u << 24); it++) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return; }
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 300000000ull) break;      // 3 s at 100 MHz
    }
    *err = 1;
}
__global__ void k_check(const double *p, size_t n, double v, unsigned *bad) { size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (k < n && p[k] != v + (double)k) atomicAdd(bad, 1u); }

struct Shm { hipIpcMemHandle_t h[2]; volatile int ready[2]; volatile int done[2]; unsigned flags[64]; };

int main() {
    const size_t n = 1 << 18;
    int fd = shm_open("/evpk_ipc_probe", O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 8192) != 0) { printf("shm failed\n"); return 2; }
    Shm *sh = (Shm *)mmap(nullptr, 8192, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    memset((void *)sh, 0, 8192);
    pid_t pid = fork();
    const int me = pid == 0 ? 1 : 0, other = 1 - me;
    CK(hipSetDevice(0));
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    // flags: the shared host page, page-locked and mapped
    CK(hipHostRegister((void *)sh, 8192, hipHostRegisterMapped | hipHostRegisterPortable));
    unsigned *dflags = nullptr;
    CK(hipHostGetDevicePointer((void **)&dflags, (void *)sh->flags, 0));
    for (int fine = 0; fine < 2; fine++) {
        double *box = nullptr;
        if (fine) { if (hipExtMallocWithFlags((void **)&box, n * 8, hipDeviceMallocFinegrained) != hipSuccess) { printf("[%d] fine-grained alloc failed\n", me); (void)hipGetLastError(); break; } }
        else CK(hipMalloc(&box, n * 8));
        CK(hipMemset(box, 0, n * 8));
        hipIpcMemHandle_t h;
        hipError_t e = hipIpcGetMemHandle(&h, box);
        if (e != hipSuccess) { printf("[%d] hipIpcGetMemHandle(fine=%d) -> %s\n", me, fine, hipGetErrorString(e)); (void)hipGetLastError(); sh->ready[me] = -1; break; }
        memcpy((void *)&sh->h[me], &h, sizeof(h));
        __sync_synchronize();
        sh->ready[me] = fine + 1;
        while (sh->ready[other] != fine + 1) { if (sh->ready[other] < 0) return 3; usleep(100); }
        hipIpcMemHandle_t ho; memcpy(&ho, (void *)&sh->h[other], sizeof(ho));
        double *peer = nullptr;
        e = hipIpcOpenMemHandle((void **)&peer, ho, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { printf("[%d] hipIpcOpenMemHandle(fine=%d) -> %s\n", me, fine, hipGetErrorString(e)); return 3; }
        unsigned *derr = nullptr; CK(hipMalloc(&derr, 8)); CK(hipMemset(derr, 0, 8));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int NIT = 200;
        CK(hipEventRecord(e0, st));
        for (int it = 1; it <= NIT; it++) {
            // write into the peer's box, signal the peer; wait for the peer's write into mine, check it
            hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, st, peer, n, (double)(1000 * it + me));
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, dflags + 16 * (fine * 2 + me), (unsigned)it);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, dflags + 16 * (fine * 2 + other), (unsigned)it, derr);
            hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, st, box, n, (double)(1000 * it + other), derr + 1);
            // the peer may overwrite my box only after I checked it: second handshake
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, dflags + 16 * (fine * 2 + me) + 8, (unsigned)it);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, dflags + 16 * (fine * 2 + other) + 8, (unsigned)it, derr);
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr[2]; CK(hipMemcpy(herr, derr, 8, hipMemcpyDeviceToHost));
        printf("[rank %d] fine=%d waitvalue_attr=%d: %d round trips of %zu KB each way: %.1f us per round trip (2 handshakes), timeouts=%u bad=%u\n",
               me, fine, can, NIT, n * 8 / 1024, 1e3 * ms / NIT, herr[0], herr[1]);
        sh->done[me] = fine + 1;
        while (sh->done[other] != fine + 1) usleep(100);
        CK(hipIpcCloseMemHandle(peer));
        CK(hipFree(box));
    }
    if (me == 0) { int stt; waitpid(pid, &stt, 0); shm_unlink("/evpk_ipc_probe"); }
    return 0;
}
