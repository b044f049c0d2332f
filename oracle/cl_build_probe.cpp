// TEST INFRASTRUCTURE: builds every .cl file named on the command line with the OpenCL runtime of this
// machine and prints the build log (the reference's BasicCL::getProgram drops it, basiccl.cpp:144-152).
// Used once to find out why the reference's initPlatform returned CL_BUILD_PROGRAM_FAILURE on the GPU box.
#define CL_TARGET_OPENCL_VERSION 220
#include <CL/cl.h>
#include <cstdio>
#include <string>
#include <vector>
int main(int argc, char **argv)
{
    cl_platform_id plat; cl_uint np = 0;
    if (clGetPlatformIDs(1, &plat, &np) != CL_SUCCESS || np == 0) { printf("no platform\n"); return 1; }
    cl_device_id dev; cl_uint nd = 0;
    cl_int e = clGetDeviceIDs(plat, CL_DEVICE_TYPE_GPU, 1, &dev, &nd);
    if (e != CL_SUCCESS || nd == 0) { printf("no gpu device (%d)\n", e); return 1; }
    cl_context ctx = clCreateContext(0, 1, &dev, NULL, NULL, &e);
    if (e != CL_SUCCESS) { printf("context %d\n", e); return 1; }
    const char *opts = getenv("CL_PROBE_OPTS");
    for (int i = 1; i < argc; i++) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) { perror(argv[i]); continue; }
        std::string src; char buf[4096]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, n);
        fclose(f);
        const char *p = src.c_str(); size_t len = src.size();
        cl_program prog = clCreateProgramWithSource(ctx, 1, &p, &len, &e);
        cl_int b = clBuildProgram(prog, 0, NULL, opts, NULL, NULL);
        size_t ls = 0;
        clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, 0, NULL, &ls);
        std::vector<char> log(ls + 1, 0);
        clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, ls, log.data(), NULL);
        printf("== %s: create %d build %d\n%s\n", argv[i], e, b, log.data());
    }
    return 0;
}
