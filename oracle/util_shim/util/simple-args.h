// TEST INFRASTRUCTURE ONLY (oracle): minimal stand-in for `geraintluff/util` simple-args.h (an empty, un-fetched git
// submodule in the reference tree, .gitmodules:1-3) exposing exactly what cmd/main.cpp:12-29 uses, so that the
// reference CLI compiles unmodified into oracle/_ref/ref_cli.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

class SimpleArgs {
	std::vector<std::string> positional;
	std::vector<std::pair<std::string, std::string>> flags; // name, value ("" when bare)
	size_t nextPositional = 0;
	bool failed = false;
	std::string message;
	template <typename T> static T parse(const std::string &s) {
		std::istringstream in(s);
		T v{};
		in >> v;
		return v;
	}
public:
	SimpleArgs(int argc, char *argv[]) {
		for (int i = 1; i < argc; ++i) {
			std::string a = argv[i];
			if (a.size() > 1 && a[0] == '-') {
				size_t start = a.find_first_not_of('-');
				std::string body = a.substr(start);
				size_t eq = body.find('=');
				if (eq == std::string::npos) flags.push_back({body, ""});
				else flags.push_back({body.substr(0, eq), body.substr(eq + 1)});
			} else {
				positional.push_back(a);
			}
		}
	}
	bool hasFlag(const std::string &name, const std::string & = "") {
		for (auto &f : flags) if (f.first == name) return true;
		return false;
	}
	template <typename T> T arg(const std::string &name, const std::string & = "") {
		if (nextPositional >= positional.size()) {
			failed = true;
			message = "missing argument: " + name;
			return T{};
		}
		return parse<T>(positional[nextPositional++]);
	}
	template <typename T> T flag(const std::string &name, const std::string &, T fallback) {
		for (auto &f : flags) if (f.first == name && !f.second.empty()) return parse<T>(f.second);
		return fallback;
	}
	void errorExit(const std::string &why = "") {
		if (!why.empty()) { std::cerr << why << "\n"; std::exit(1); }
		if (failed) { std::cerr << message << "\n"; std::exit(1); }
	}
};
template <> inline std::string SimpleArgs::parse<std::string>(const std::string &s) { return s; }
