#include "db_host.h"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace oar::host;
int main(){
  std::vector<uint8_t> m(960*960); FILE*f=fopen("gpurun_out/mask.bin","rb"); fread(m.data(),1,m.size(),f); fclose(f);
  auto t0=std::chrono::steady_clock::now();
  std::vector<Contour> cs; for(int r=0;r<20;++r) cs=find_contours(m.data(),960,960,1000);
  auto t1=std::chrono::steady_clock::now();
  int nb=0; for(int r=0;r<20;++r){ nb=0; for(auto&c:cs){ auto s=simplify_chain(c.pts); Pt mb[4]; float ms; bool ok = s.size()>=3? mini_box(s,mb,ms):mini_box(c.pts,mb,ms); if(ok&&ms>=3) nb++; } }
  auto t2=std::chrono::steady_clock::now();
  printf("contours=%zu boxes=%d find=%.3f ms minibox=%.3f ms\n", cs.size(), nb, std::chrono::duration<double,std::milli>(t1-t0).count()/20, std::chrono::duration<double,std::milli>(t2-t1).count()/20);
}
