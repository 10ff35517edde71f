#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e));}}while(0)
static void show(const char*name, void*p, size_t n){
  int mode=-1; hipError_t e=hipMemRangeGetAttribute(&mode,sizeof mode,hipMemRangeAttributeCoherencyMode,p,n);
  hipPointerAttribute_t a; hipError_t e2=hipPointerGetAttributes(&a,p);
  printf("%s: coherency query %s mode %d (fine=%d coarse=%d indeterminate=%d); pointer attrs %s type %d device %d allocFlags %u\n",name,hipGetErrorString(e),mode,
    hipMemRangeCoherencyModeFineGrain,hipMemRangeCoherencyModeCoarseGrain,hipMemRangeCoherencyModeIndeterminate,hipGetErrorString(e2),(int)a.type,a.device,a.allocationFlags);
}
int main(){
  void*p=nullptr; CK(hipMalloc(&p,8<<20)); show("hipMalloc",p,8<<20);
  void*q=nullptr; CK(hipExtMallocWithFlags(&q,8<<20,hipDeviceMallocFinegrained)); show("finegrained",q,8<<20);
  hipMemAllocationProp prop={}; prop.type=hipMemAllocationTypePinned; prop.location.type=hipMemLocationTypeDevice; prop.location.id=0;
  size_t g=0; CK(hipMemGetAllocationGranularity(&g,&prop,hipMemAllocationGranularityRecommended));
  size_t gm=0; CK(hipMemGetAllocationGranularity(&gm,&prop,hipMemAllocationGranularityMinimum));
  printf("granularity recommended %zu minimum %zu\n",g,gm);
  char*va=nullptr; CK(hipMemAddressReserve((void**)&va,4*g,0,nullptr,0));
  hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h,4*g,&prop,0)); CK(hipMemMap(va,4*g,0,h,0));
  hipMemAccessDesc acc={}; acc.location=prop.location; acc.flags=hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess(va,4*g,&acc,1));
  show("vmm",va,4*g);
  return 0;
}
