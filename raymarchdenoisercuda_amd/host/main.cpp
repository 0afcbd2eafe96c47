// main.cpp — command line of the host harness.  It honours the contract of the reference's
// src/main.cpp:5-40 (`-t [label]` runs the registered tests whose name matches the regex label, all of
// them when the label is omitted; `-h` prints the help and exits with 1; no arguments = help), and adds
// what a build check needs: several `-t` in one call and an exit status that reports failed tests.
#include <cstdio>
#include <string>
#include <vector>

#include "test.h"

namespace {

struct Request {
    std::vector<std::string> labels;     // one regex per -t (".*" when -t came without a label)
    bool help = false;
    std::vector<std::string> unknown;
};

Request parse(const std::vector<std::string>& args)
{
    Request r;
    for (size_t k = 0; k < args.size(); ++k) {
        const std::string& a = args[k];
        if (a == "-h") {
            r.help = true;
        } else if (a == "-t") {
            const bool has_label = k + 1 < args.size() && !args[k + 1].empty() && args[k + 1][0] != '-';
            r.labels.push_back(has_label ? args[++k] : std::string(".*"));
        } else {
            r.unknown.push_back(a);
        }
    }
    return r;
}

int help(const char* prog)
{
    std::printf("Usage: %s [options]\n"
                "Options:\n"
                "  -t [label]   Run tests (all or specific label)\n"
                "  -h           Show this help message\n"
                "Exit status: 0 all selected tests passed, 1 help shown, 2 a test failed.\n", prog);
    return 1;
}

}  // namespace

int main(int argc, char* argv[])
{
    const Request req = parse(std::vector<std::string>(argv + 1, argv + argc));
    for (const std::string& u : req.unknown) std::fprintf(stderr, "Unknown option: %s\n", u.c_str());
    if (req.help || (req.labels.empty() && req.unknown.empty())) return help(argv[0]);
    for (const std::string& label : req.labels) test(label);
    return testFailures() ? 2 : 0;
}
