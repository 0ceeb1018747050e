// Minimal JSON document reader for scene files.
//
// The scene format keeps every scalar as a JSON *string* (the reference parses them
// with istringstream, scene.cpp:25-59), so this reader only has to deliver objects,
// arrays and strings faithfully; numbers, booleans and null are accepted and kept as
// their source text / type tag.  Object members keep file order; lookup is linear
// (scene files have a handful of keys per object).
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rtxhost {

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    std::string text;                                   // String / Number / Bool payload
    std::vector<Json> items;                            // Array
    std::vector<std::pair<std::string, Json>> members;  // Object

    bool is_null() const { return type == Null; }
    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
    bool is_string() const { return type == String; }

    bool contains(const std::string& key) const {
        if (type != Object) return false;
        for (const auto& m : members) if (m.first == key) return true;
        return false;
    }
    // Missing keys yield a shared null value (nlohmann's operator[] on a const object
    // would assert; scene files written for the reference never rely on that).
    const Json& operator[](const std::string& key) const {
        static const Json null_value;
        if (type != Object) return null_value;
        for (const auto& m : members) if (m.first == key) return m.second;
        return null_value;
    }
    // True when the key exists and is not JSON null: the guard the reference uses
    // everywhere (`contains(k) && !j[k].is_null()`).
    bool has(const std::string& key) const { return contains(key) && !(*this)[key].is_null(); }

    const std::string& str() const {
        if (type != String) throw std::runtime_error("JSON value is not a string");
        return text;
    }

    static Json parse(const std::string& src) {
        Parser p{src, 0};
        Json v = p.value();
        p.ws();
        if (p.pos != src.size()) p.fail("trailing characters");
        return v;
    }

private:
    struct Parser {
        const std::string& s;
        size_t pos;
        [[noreturn]] void fail(const char* what) const {
            throw std::runtime_error(std::string("JSON parse error at byte ") +
                                     std::to_string(pos) + ": " + what);
        }
        void ws() {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) pos++;
        }
        static void utf8(std::string& out, uint32_t cp) {
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) {
                out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F));
                out += (char)(0x80 | (cp & 0x3F));
            } else {
                out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F));
                out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
            }
        }
        uint32_t hex4() {
            if (pos + 4 > s.size()) fail("short \\u escape");
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) {
                char c = s[pos++];
                v <<= 4;
                if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
                else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
                else fail("bad \\u escape");
            }
            return v;
        }
        std::string string_body() {
            std::string out;
            if (s[pos] != '"') fail("expected string");
            pos++;
            // Fast path: copy runs without escapes (vertex / face payloads are megabytes).
            while (true) {
                size_t start = pos;
                while (pos < s.size() && s[pos] != '"' && s[pos] != '\\') pos++;
                out.append(s, start, pos - start);
                if (pos >= s.size()) fail("unterminated string");
                if (s[pos] == '"') { pos++; return out; }
                pos++;  // backslash
                if (pos >= s.size()) fail("unterminated escape");
                char e = s[pos++];
                switch (e) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        uint32_t cp = hex4();
                        if (cp >= 0xD800 && cp < 0xDC00 && pos + 1 < s.size() && s[pos] == '\\' && s[pos + 1] == 'u') {
                            pos += 2;
                            uint32_t lo = hex4();
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(out, cp);
                        break;
                    }
                    default: fail("bad escape");
                }
            }
        }
        Json value() {
            ws();
            if (pos >= s.size()) fail("unexpected end");
            Json v;
            char c = s[pos];
            if (c == '{') {
                v.type = Object;
                pos++;
                ws();
                if (pos < s.size() && s[pos] == '}') { pos++; return v; }
                while (true) {
                    ws();
                    std::string key = string_body();
                    ws();
                    if (pos >= s.size() || s[pos] != ':') fail("expected ':'");
                    pos++;
                    Json child = value();
                    // duplicate keys: the last one wins, as in nlohmann's map-backed object
                    bool replaced = false;
                    for (auto& m : v.members) if (m.first == key) { m.second = std::move(child); replaced = true; break; }
                    if (!replaced) v.members.emplace_back(std::move(key), std::move(child));
                    ws();
                    if (pos < s.size() && s[pos] == ',') { pos++; continue; }
                    if (pos < s.size() && s[pos] == '}') { pos++; return v; }
                    fail("expected ',' or '}'");
                }
            }
            if (c == '[') {
                v.type = Array;
                pos++;
                ws();
                if (pos < s.size() && s[pos] == ']') { pos++; return v; }
                while (true) {
                    v.items.push_back(value());
                    ws();
                    if (pos < s.size() && s[pos] == ',') { pos++; continue; }
                    if (pos < s.size() && s[pos] == ']') { pos++; return v; }
                    fail("expected ',' or ']'");
                }
            }
            if (c == '"') { v.type = String; v.text = string_body(); return v; }
            if (s.compare(pos, 4, "null") == 0) { pos += 4; v.type = Null; return v; }
            if (s.compare(pos, 4, "true") == 0) { pos += 4; v.type = Bool; v.text = "true"; return v; }
            if (s.compare(pos, 5, "false") == 0) { pos += 5; v.type = Bool; v.text = "false"; return v; }
            size_t start = pos;
            while (pos < s.size() && (s[pos] == '-' || s[pos] == '+' || s[pos] == '.' || s[pos] == 'e' ||
                                      s[pos] == 'E' || (s[pos] >= '0' && s[pos] <= '9'))) pos++;
            if (pos == start) fail("unexpected character");
            v.type = Number;
            v.text = s.substr(start, pos - start);
            return v;
        }
    };
};

}  // namespace rtxhost
