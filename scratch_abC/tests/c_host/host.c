/* TEST INFRASTRUCTURE: a plain-C99 host of the C ABI (include/lbc_hip.h) -- what a cgo / JNI / FFI binding would do.
 * Loads the library given as argv[1], plans ImagePolicyModelSS('resnet34') at 160x384 and walks its tensor table
 * (host-side calls only: no device memory, no launches), then prints one line the test parses. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "lbc_hip.h"

typedef int (*create_fn)(const lbc_net_desc*, lbc_net**);
typedef void (*destroy_fn)(lbc_net*);
typedef int (*count_fn)(const lbc_net*);
typedef int (*info_fn)(const lbc_net*, int, char*, int, int*, int*, int*);
typedef size_t (*ws_fn)(const lbc_net*);
typedef const char* (*str_fn)(void);
typedef int (*stages_fn)(void);
typedef size_t (*desc_ws_fn)(const lbc_conv_desc*);

int main(int argc, char** argv)
{
    void* h;
    lbc_net_desc d;
    lbc_net* net = NULL;
    int n, i, params = 0, buffers = 0;
    long long elems = 0;
    char first[256] = "", last[256] = "", name[256];
    if (argc < 2) return 2;
    h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 3; }
    /* the library must be the ABI this header describes */
    if (((stages_fn)dlsym(h, "lbc_version"))() != LBC_HIP_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", ((stages_fn)dlsym(h, "lbc_version"))(), LBC_HIP_ABI_VERSION); return 7; }
    {
        /* descriptors: one initialised as the header says is accepted (a size query: no device work), a zeroed / stale one refused */
        lbc_conv_desc c = LBC_CONV_DESC_INIT, stale;
        c.N = 2; c.H = 8; c.W = 8; c.C = 64; c.K = 64; c.KH = 3; c.KW = 3; c.S = 1; c.P = 1;
        if (c.struct_size != sizeof(lbc_conv_desc) || c.split_workspace != NULL || c.relu != 0) return 8;
        if (((desc_ws_fn)dlsym(h, "lbc_conv2d_wgrad_workspace"))(&c) == 0) return 9;
        memset(&stale, 0, sizeof(stale));
        stale.N = 2; stale.H = 8; stale.W = 8; stale.C = 64; stale.K = 64; stale.KH = 3; stale.KW = 3; stale.S = 1; stale.P = 1;
        if (((desc_ws_fn)dlsym(h, "lbc_conv2d_wgrad_workspace"))(&stale) != 0) return 10;
        if (!strstr(((str_fn)dlsym(h, "lbc_last_error"))(), "struct_size")) return 11;
    }
    memset(&d, 0, sizeof(d));
    d.arch = 34; d.in_channels = 3; d.H = 160; d.W = 384; d.normalize = 1; d.max_batch = 2; d.precision = 0;
    if (((create_fn)dlsym(h, "lbc_net_create"))(&d, &net) != 0) {
        fprintf(stderr, "create: %s\n", ((str_fn)dlsym(h, "lbc_last_error"))());
        return 4;
    }
    n = ((count_fn)dlsym(h, "lbc_net_num_tensors"))(net);
    for (i = 0; i < n; ++i) {
        int kind = -1, ndim = 0, shape[4] = {1, 1, 1, 1}, k;
        long long e = 1;
        if (((info_fn)dlsym(h, "lbc_net_tensor_info"))(net, i, name, (int)sizeof(name), &kind, &ndim, shape) != 0) return 5;
        for (k = 0; k < ndim; ++k) e *= shape[k];
        if (kind == 0) { ++params; elems += e; } else ++buffers;
        if (i == 0) strcpy(first, name);
        strcpy(last, name);
    }
    /* a bad plan must come back as an error code + message, not a crash */
    {
        lbc_net* bad = NULL;
        lbc_net_desc b = d;
        b.arch = 50;
        if (((create_fn)dlsym(h, "lbc_net_create"))(&b, &bad) == 0) return 6;
    }
    printf("backend=%s tensors=%d params=%d buffers=%d param_elems=%lld first=%s last=%s workspace=%zu stages=%d\n",
           ((str_fn)dlsym(h, "lbc_backend"))(), n, params, buffers, elems, first, last,
           ((ws_fn)dlsym(h, "lbc_net_workspace_bytes"))(net), ((stages_fn)dlsym(h, "lbc_net_num_stages"))());
    ((destroy_fn)dlsym(h, "lbc_net_destroy"))(net);
    return 0;
}
