// test.h — the reference's tiny test registry (include/test.h:6-22): TEST(name) registers a body,
// SKIP(name) defines it without registering, test(wildcard) runs the bodies whose name matches
// the regex and prints "Passed with %.3f ms" / "Fail with ...".
#ifndef RMD_TEST_H
#define RMD_TEST_H

#include <functional>
#include <string>
#include <utility>
#include <vector>

typedef std::vector<std::pair<std::string, std::function<void()>>> FuncVector;

FuncVector& registeredFuncs();   // function-local static: registration order no longer depends on TU layout

#define TEST(func_name)                                                          \
    void func_name();                                                            \
    static struct func_name##_registrar {                                        \
        func_name##_registrar() { registeredFuncs().push_back({#func_name, func_name}); } \
    } func_name##_instance;                                                      \
    void func_name()

#define SKIP(func_name) void func_name()

// Throws std::runtime_error("<what>") when cond is false: a real assertion channel, which the
// reference's tests lack.
void expect(bool cond, const std::string& what);

void test(std::string wildcard = ".*");   // reference include/test.h:22: returns nothing
int testFailures();                       // bodies that threw since the process started (the CLI's exit status)

#endif
