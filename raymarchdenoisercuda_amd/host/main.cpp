// main.cpp — same CLI contract as the reference's src/main.cpp:5-40: `-t [label]` runs the
// registered tests whose name matches the regex label (all when omitted), `-h` prints help.
// Unlike the reference the exit status reports failed tests.
#include <cstring>
#include <iostream>

#include "test.h"

static void usage(const char* prog)
{
    std::cout << "Usage: " << prog << " [options]\n"
              << "Options:\n"
              << "  -t [label]   Run tests (all or specific label)\n"
              << "  -h           Show this help message\n";
}

int main(int argc, char* argv[])
{
    if (argc < 2) { usage(argv[0]); return 1; }
    int failed = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-t")) {
            if (i + 1 < argc && argv[i + 1][0] != '-') failed += test(argv[++i]);
            else failed += test();
        } else if (!strcmp(argv[i], "-h")) {
            usage(argv[0]);
            return 1;
        } else {
            std::cerr << "Unknown option: " << argv[i] << std::endl;
        }
    }
    return failed ? 2 : 0;
}
