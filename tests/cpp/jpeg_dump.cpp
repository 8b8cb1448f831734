// jpeg_dump.cpp — decodes JPEG files with the C++ host mirror's decoder (include/utopian_jpeg.hpp) for tests/test_jpeg.py.
//   usage: jpeg_dump <list.txt>   where each line is "<in.jpg> <out.raw>"; per line prints "ok W H C progressive" or "error <message>".
// The raw file holds W*H*C bytes. Built with ASan + UBSan by the test.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <sstream>

#include "utopian_jpeg.hpp"

int main(int argc, char** argv) {
   if (argc < 2) return 2;
   std::ifstream list(argv[1]);
   std::string line;
   while (std::getline(list, line)) {
      std::istringstream ss(line);
      std::string in, out;
      if (!(ss >> in >> out)) continue;
      std::ifstream f(in, std::ios::binary);
      std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      try {
         utopian::jpeg::Image img = utopian::jpeg::decode(d);
         std::ofstream o(out, std::ios::binary);
         o.write(reinterpret_cast<const char*>(img.pixels.data()), (std::streamsize)img.pixels.size());
         std::printf("ok %u %u %u %d\n", img.width, img.height, img.channels, (int)img.progressive);
      } catch (const utopian::Error& e) {
         std::printf("error %s\n", e.what());
      }
   }
   return 0;
}
